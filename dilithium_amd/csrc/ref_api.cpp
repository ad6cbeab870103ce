// ref_api.cpp -- libdil256_ref.so: the reference-identical C++ signatures (include/dil256_ref.hpp)
// as thin batch-of-one wrappers over the C-ABI.  No arithmetic happens here: every function
// forwards to a HIP kernel through include/dil256.h and aborts loudly if that fails (the
// reference's functions are void and cannot report errors -- SURVEY 8b).
#include "../../include/dil256.h"
#include "../../include/dil256_ref.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

[[noreturn]] void die(const char* what, int rc)
{
    fprintf(stderr, "libdil256_ref: %s failed: hipError %d (%s) -- no CPU fallback\n", what, rc, dil_error_string(rc));
    abort();
}
inline void ok(const char* what, int rc)
{
    if (rc) die(what, rc);
}

// Every function of this library is a batch of one: serve them from the resident mailbox wave (include/dil256.h "HOST MAILBOX")
// instead of a launch per call, unless the environment says otherwise.
struct MailboxOn {
    MailboxOn()
    {
        if (!getenv("DIL_HOST_MAILBOX")) (void)dil_set_option("host_mailbox", 1);
    }
} mailbox_on;

// zeta^brv8(k), zeta = 1753, centred -- the table of consts.cpp:64-97, computed at compile time
constexpr int64_t Qc = DILITHIUM_Q;
constexpr unsigned brv8(unsigned x)
{
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r |= ((x >> i) & 1u) << (7 - i);
    return r;
}
constexpr int64_t powmod(int64_t b, unsigned e)
{
    int64_t r = 1;
    b %= Qc;
    while (e) {
        if (e & 1) r = r * b % Qc;
        b = b * b % Qc;
        e >>= 1;
    }
    return r;
}
struct ZetaTable {
    data_t v[DILITHIUM_N];
    constexpr ZetaTable() : v()
    {
        v[0] = 0;
        for (unsigned k = 1; k < DILITHIUM_N; k++) {
            int64_t z = powmod(1753, brv8(k));
            v[k] = (data_t)(z > (Qc - 1) / 2 ? z - Qc : z);
        }
    }
};
constexpr ZetaTable ZT{};

}  // namespace

#define Z(i) ZT.v[i]
#define Z8(i) Z(i), Z(i + 1), Z(i + 2), Z(i + 3), Z(i + 4), Z(i + 5), Z(i + 6), Z(i + 7)
#define Z64(i) Z8(i), Z8(i + 8), Z8(i + 16), Z8(i + 24), Z8(i + 32), Z8(i + 40), Z8(i + 48), Z8(i + 56)
extern const data_t zetas_barrett[DILITHIUM_N] = {Z64(0), Z64(64), Z64(128), Z64(192)};

// consts_hw.cpp:2-91: the twiddle ROM of the radix-2x2 butterfly unit, row t = (z[k], z[2k], z[2k+1]) for
// k = 1; 4..7; 16..31; 64..127 (one row per butterfly group of passes 0..3).  Our kernels do not read it -- it is
// exported because it is part of the reference's data surface (consts_hw.h:7).
#define H(k) {Z(k), Z(2 * (k)), Z(2 * (k) + 1)}
#define H4(k) H(k), H(k + 1), H(k + 2), H(k + 3)
#define H16(k) H4(k), H4(k + 4), H4(k + 8), H4(k + 12)
extern const data_t zetas_barrett_hw[85][3] = {H(1), H4(4), H16(16), H16(64), H16(80), H16(96), H16(112)};
#undef H16
#undef H4
#undef H

void ntt(data_t a[DILITHIUM_N]) { ok("ntt", dil_ntt_host(a, 1)); }
void invntt(data_t a[DILITHIUM_N]) { ok("invntt", dil_invntt_host(a, 1)); }
void pointwise_barrett(data_t c[DILITHIUM_N], const data_t a[DILITHIUM_N], const data_t b[DILITHIUM_N])
{
    ok("pointwise_barrett", dil_pointwise_host(c, a, b, 1));
}
void ntt2x2_ref(data_t a[DILITHIUM_N]) { ok("ntt2x2_ref", dil_ntt_host(a, 1)); }
void invntt2x2_ref(data_t a[DILITHIUM_N]) { ok("invntt2x2_ref", dil_invntt_host(a, 1)); }

// the `mode` argument only selects the datapath inside each reference function
// (ntt2x2_fwdntt.cpp:122-129); each entry point has exactly one meaningful mode
void ntt2x2_fwdntt(bram* ram, enum OPERATION, enum MAPPING mapping)
{
    ok("ntt2x2_fwdntt", dil_bram_fwdntt_host(&ram->coeffs[0][0], 1, (int)mapping));
}
void ntt2x2_invntt(bram* ram, enum OPERATION, enum MAPPING mapping)
{
    ok("ntt2x2_invntt", dil_bram_invntt_host(&ram->coeffs[0][0], 1, (int)mapping));
}
void ntt2x2_mul(bram* ram, const bram* mul_ram, enum MAPPING mapping)
{
    ok("ntt2x2_mul", dil_bram_mul_host(&ram->coeffs[0][0], &mul_ram->coeffs[0][0], 1, (int)mapping));
}

unsigned resolve_address(enum MAPPING mapping, unsigned addr)   // address_encoder_decoder.cpp:34-55
{
    switch (mapping) {
    case AFTER_INVNTT: return (addr % 16) * 4 + addr / 16;
    case AFTER_NTT: return (addr % 4) * 16 + addr / 4;
    default: return addr;
    }
}

void reshape(bram* ram, const data_t in[DILITHIUM_N])           // util.cpp:61-72: row i = coefficients 4i..4i+3
{
    for (int i = 0; i < BRAM_DEPT; i++)
        for (int j = 0; j < 4; j++) ram->coeffs[i][j] = in[4 * i + j];
}

// ---- the rest of the hardware model's helper surface, so that the reference's own
// ntt2x2_test.cpp links against this library with nothing else on the link line ---------------------------
// ram_util.h:29-31: one `bram` row = 4 coefficients
void read_ram(data_t data_out[4], const bram* ram, const unsigned ram_i) { memcpy(data_out, ram->coeffs[ram_i], 4 * sizeof(data_t)); }
void write_ram(bram* ram, const unsigned ram_i, const data_t data_in[4]) { memcpy(ram->coeffs[ram_i], data_in, 4 * sizeof(data_t)); }

// ram_util.h:33 -- the twiddle ROM address generator of the butterfly unit (ram_util.cpp:45-94 ==
// twiddle_resolver.v:87-130): group `i` of pass `level` (0, 2, 4, 6) reads ROM row base(level') + (i' mod 4^(level'/2))
// with base = 0, 1, 5, 21; forward: level' = level, i' = i, outputs (z[k], z[k], z[2k], z[2k+1]);
// inverse: level' = 6 - level, i' = 63 - i, outputs (z[2k+1], z[2k], z[k], z[k]).
void get_twiddle_factors(data_t data_out[4], int i, int level, OPERATION mode)
{
    static const unsigned row_base[4] = {0, 1, 5, 21};
    if (mode != FORWARD_NTT_MODE && mode != INVERSE_NTT_MODE) {      // the reference reads row 0, column 0 four times
        data_out[0] = data_out[1] = data_out[2] = data_out[3] = zetas_barrett_hw[0][0];
        return;
    }
    const bool fwd = mode == FORWARD_NTT_MODE;
    const int lv = fwd ? level : DILITHIUM_LOGN - 2 - level;
    const unsigned grp = (unsigned)(fwd ? i : BRAM_DEPT - 1 - i) & ((1u << lv) - 1u);
    const data_t* row = zetas_barrett_hw[row_base[lv >> 1] + grp];
    if (fwd) {
        data_out[0] = data_out[1] = row[0];
        data_out[2] = row[1];
        data_out[3] = row[2];
    } else {
        data_out[0] = row[2];
        data_out[1] = row[1];
        data_out[2] = data_out[3] = row[0];
    }
}

// util.h:46: 0 = equal (exact comparison, as the reference's)
int compare_array(data_t* a, data_t* b, int bound) { return memcmp(a, b, (size_t)bound * sizeof(data_t)) != 0; }

// util.h:48-50: `array` (reference order) against `ram` behind `mapping`, both reduced to [0, q) first; prints the
// first mismatching row in the reference's format and returns 1, else 0
int compare_bram_array(bram* ram, data_t array[DILITHIUM_N], const char* string, enum MAPPING mapping, int print_out)
{
    auto canon = [](data_t v) { return (data_t)(((int64_t)v % Qc + Qc) % Qc); };
    for (int r = 0; r < BRAM_DEPT; r++) {
        const unsigned addr = resolve_address(mapping, (unsigned)r);
        data_t gold[4], got[4];
        bool same = true;
        for (int j = 0; j < 4; j++) {
            gold[j] = canon(array[4 * r + j]);
            got[j] = canon(ram->coeffs[addr][j]);
            same = same && gold[j] == got[j];
        }
        if (print_out) {
            printf("%d: %d, %d, %d, %d\n", r, gold[0], gold[1], gold[2], gold[3]);
            printf("[%d]: |%d, %d, %d, %d|\n--------------\n", 4 * r, got[0], got[1], got[2], got[3]);
        }
        if (!same) {
            printf("%s Error at index: %d => %u\n", string, 4 * r, addr);
            printf("gold: %12u | %12u | %12u | %12u [*]\n", gold[0], gold[1], gold[2], gold[3]);
            printf("test: %12u | %12u | %12u | %12u\n", got[0], got[1], got[2], got[3]);
            return 1;
        }
    }
    return 0;
}

void print_reshaped_array(bram* ram, int bound, const char* string)    // util.h:40
{
    printf("%s :\n", string);
    for (int i = 0; i < bound; i++)
        for (int j = 0; j < 4; j++) printf("%u, ", ram->coeffs[i][j]);
    printf("\n");
}
void print_index_reshaped_array(bram* ram, int index)                  // util.h:42
{
    printf("[%d]: ", index);
    for (int j = 0; j < 4; j++) printf("%u, ", ram->coeffs[index][j]);
    printf("\n");
}
