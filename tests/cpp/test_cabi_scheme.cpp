// A C++ host program on the C-ABI (include/dil256.h) alone -- no Python, no torch: the shape of host code north_star asks
// for.  Key generation, signing and verification of ragged MESSAGES (mu is hashed on the device) through the host-buffer
// and device-pointer entry points and through the multi-GPU host layer; every result is cross-checked against another
// path of the library (the byte-exact checks against the reference's KAT files live in tests/test_gpu_codecs.py and
// tests/test_gpu_msg.py, which drive the same entry points).
//   usage: test_cabi_scheme [level] [batch]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/dil256.h"

#define CK(call)                                                                         \
    do {                                                                                 \
        const int rc__ = (int)(call);                                                    \
        if (rc__) {                                                                      \
            printf("%s failed: %d (%s)\nERROR\n", #call, rc__, dil_error_string(rc__));  \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char** argv)
{
    const int level = argc > 1 ? atoi(argv[1]) : 3;
    const size_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 300;
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    if (!pkb) return printf("bad level\nERROR\n"), 1;
    CK(dil_init(0));
    srand(7);
    // ---- one key pair from a seed, host buffers ------------------------------------------------------------
    std::vector<uint8_t> seed(32), pk(pkb), sk(skb);
    for (auto& b : seed) b = (uint8_t)rand();
    CK(dil_keygen_host(pk.data(), sk.data(), seed.data(), level, 1));
    if (memcmp(pk.data(), sk.data(), 32) != 0) return printf("rho of pk and sk differ\nERROR\n"), 1;
    // ---- ragged messages on the device: blob + offsets + lengths ---------------------------------------------
    std::vector<uint64_t> offs(n);
    std::vector<uint32_t> lens(n);
    std::vector<uint8_t> blob;
    for (size_t i = 0; i < n; i++) {
        offs[i] = blob.size();
        lens[i] = (uint32_t)(rand() % 700);
        for (uint32_t j = 0; j < lens[i]; j++) blob.push_back((uint8_t)rand());
    }
    blob.push_back(0);
    uint8_t *d_blob, *d_sk, *d_pk, *d_sig, *d_mu;
    uint64_t* d_offs;
    uint32_t* d_lens;
    int32_t *d_att, *d_verdict;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipMalloc(&d_blob, blob.size()));
    CK(hipMalloc(&d_offs, n * 8));
    CK(hipMalloc(&d_lens, n * 4));
    CK(hipMalloc(&d_sk, skb));
    CK(hipMalloc(&d_pk, pkb));
    CK(hipMalloc(&d_sig, n * sgb));
    CK(hipMalloc(&d_mu, n * 64));
    CK(hipMalloc(&d_att, n * 4));
    CK(hipMalloc(&d_verdict, n * 4));
    CK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_offs, offs.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lens, lens.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sk, sk.data(), skb, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pk, pk.data(), pkb, hipMemcpyHostToDevice));
    // sign (sk, M) and verify (pk, M, sig) on the caller's stream
    CK(dil_sign_msg_dev(d_sig, d_att, d_sk, d_blob, blob.size(), d_offs, d_lens, level, n, /*shared_sk*/ 1, 512, st));
    CK(dil_verify_msg_dev(d_verdict, d_pk, d_sig, d_blob, blob.size(), d_offs, d_lens, level, n, /*shared_pk*/ 1, st));
    std::vector<int32_t> verdict(n), att(n);
    std::vector<uint8_t> sig(n * sgb), mu(n * 64);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(verdict.data(), d_verdict, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(att.data(), d_att, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sig.data(), d_sig, n * sgb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (verdict[i] != 0 || att[i] < 1) return printf("message %zu: verdict %d attempts %d\nERROR\n", i, verdict[i], att[i]), 1;
    // ---- the same signatures from the digest path: mu on the device, then the host-buffer multi-GPU layer -------
    CK(dil_mu_dev(d_mu, d_sk + 64, 0, d_blob, blob.size(), d_offs, d_lens, /*bad*/ nullptr, n, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(mu.data(), d_mu, n * 64, hipMemcpyDeviceToHost));
    std::vector<uint8_t> sig2(n * sgb);
    std::vector<int32_t> att2(n), v2(n);
    CK(dil_sign_multi_host(sig2.data(), att2.data(), sk.data(), mu.data(), level, n, 1, 512, /*ndev: all*/ 0));
    if (sig != sig2 || att != att2) return printf("sign_msg_dev and sign_multi_host disagree\nERROR\n"), 1;
    CK(dil_verify_sig_multi_host(v2.data(), pk.data(), sig2.data(), mu.data(), level, n, 1, 0));
    for (size_t i = 0; i < n; i++)
        if (v2[i] != 0) return printf("multi-host verify rejected %zu\nERROR\n", i), 1;
    // ---- tampering is caught: a message byte, a signature byte --------------------------------------------------
    if (lens[1] > 0) blob[offs[1]] ^= 1;
    sig2[2 * sgb + 40] ^= 4;
    CK(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sig, sig2.data(), n * sgb, hipMemcpyHostToDevice));
    CK(dil_verify_msg_dev(d_verdict, d_pk, d_sig, d_blob, blob.size(), d_offs, d_lens, level, n, 1, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(verdict.data(), d_verdict, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        const bool bad = (i == 1 && lens[1] > 0) || i == 2;
        if ((verdict[i] != 0) != bad) return printf("tamper check: message %zu verdict %d\nERROR\n", i, verdict[i]), 1;
    }
    // ---- options ---------------------------------------------------------------------------------------------
    int v = -1;
    CK(dil_set_option("zeroize", 1));
    CK(dil_get_option("zeroize", &v));
    if (v != 1 || dil_set_option("no_such_option", 1) == 0) return printf("options\nERROR\n"), 1;
    CK(dil_sign_msg_dev(d_sig, d_att, d_sk, d_blob, blob.size(), d_offs, d_lens, level, n, 1, 512, st));      // with scratch wiping
    CK(hipStreamSynchronize(st));
    CK(dil_set_option("zeroize", 0));
    CK(dil_shutdown());
    printf("level %d, %zu messages: keygen / sign_msg / verify_msg / mu / multi-host layer agree\nOK\n", level, n);
    return 0;
}
