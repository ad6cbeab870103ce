// dil256_ref.hpp -- the reference's dilithium-256/ C++ surface, re-declared for the drop-in.
//
// A translation unit written against the reference's headers
//     params.h:30-35        data_t, data2_t, DILITHIUM_Q / _N / _LOGN
//     consts.h:30           extern const data_t zetas_barrett[DILITHIUM_N]
//     ref_ntt.h:30-36       ntt, pointwise_barrett, invntt
//     ref_ntt2x2.h:31-33    ntt2x2_ref, invntt2x2_ref
//     config.h:29-51        BRAM<T>, bram, enum OPERATION, enum MAPPING
//     ntt2x2.h:30-34        ntt2x2_fwdntt, ntt2x2_mul, ntt2x2_invntt
//     address_encoder_decoder.h   resolve_address
//     util.h:31-50          print_array<T>, reshape, compare_array, compare_bram_array, print_reshaped_array, print_index_reshaped_array
//     ram_util.h:29-33      read_ram, write_ram, get_twiddle_factors
// compiles against this header and links against libdil256_ref.so + libdil256.so instead:
// same names, same C++ linkage, same in-place / caller-owns-buffers contract.  Every call runs
// on the GPU (batch = 1 through the host-pointer C-ABI of include/dil256.h); results are the
// canonical residues in [0, q) -- congruent mod q to what the reference returns, which is the
// equality its own tests use (ref_test_ntt_ntt2x2.cpp:31-42, util.cpp:98-112).  For throughput
// use the batched entry points of dil256.h; this header exists for source compatibility.
#ifndef DIL256_REF_HPP
#define DIL256_REF_HPP

#include <stdint.h>
#include <stdio.h>

typedef int32_t data_t;
typedef int64_t data2_t;

#define DILITHIUM_Q 8380417
#define DILITHIUM_N 256
#define DILITHIUM_LOGN 8
#define BRAM_DEPT (DILITHIUM_N / 4)

template <typename T>
struct BRAM {
    T coeffs[BRAM_DEPT][4];
};
typedef BRAM<data_t> bram;

enum OPERATION { FORWARD_NTT_MODE, INVERSE_NTT_MODE, MUL_MODE };
enum MAPPING { NATURAL, AFTER_NTT, AFTER_INVNTT };

extern const data_t zetas_barrett[DILITHIUM_N];
extern const data_t zetas_barrett_hw[85][3];   // consts_hw.h:7

void ntt(data_t a[DILITHIUM_N]);
void invntt(data_t a[DILITHIUM_N]);
void pointwise_barrett(data_t c[DILITHIUM_N], const data_t a[DILITHIUM_N], const data_t b[DILITHIUM_N]);

void ntt2x2_ref(data_t a[DILITHIUM_N]);
void invntt2x2_ref(data_t a[DILITHIUM_N]);

void ntt2x2_fwdntt(bram* ram, enum OPERATION mode, enum MAPPING mapping);
void ntt2x2_mul(bram* ram, const bram* mul_ram, enum MAPPING mapping);
void ntt2x2_invntt(bram* ram, enum OPERATION mode, enum MAPPING mapping);

unsigned resolve_address(enum MAPPING mapping, unsigned addr);
void reshape(bram* ram, const data_t in[DILITHIUM_N]);
int compare_array(data_t* a, data_t* b, int bound);
int compare_bram_array(bram* ram, data_t array[DILITHIUM_N], const char* string, enum MAPPING mapping, int print_out);
void print_reshaped_array(bram* ram, int bound, const char* string);
void print_index_reshaped_array(bram* ram, int index);
// util.h:31-40 is a header template there, so it is one here: "<label> :" and the first `bound` entries as "%3u, " on one line
// (the report format is the surface; no arithmetic)
template <typename T>
void print_array(T* a, int bound, const char* string)
{
    fputs(string, stdout);
    fputs(" :", stdout);
    for (T* p = a; p != a + bound; ++p) printf("%3u, ", *p);
    putchar('\n');
}

void read_ram(data_t data_out[4], const bram* ram, const unsigned ram_i);
void write_ram(bram* ram, const unsigned ram_i, const data_t data_in[4]);
void get_twiddle_factors(data_t data_out[4], int i, int level, OPERATION mode);

#endif
