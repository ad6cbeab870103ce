// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE (part of the checker oracle/_ref/libref.so, never shipped, never linked by the product).
// The reference keeps its 2x2 butterfly unit as header TEMPLATES (dilithium-256/hardware_code/butterfly_unit.h:29-110 `butterfly`,
// :112-196 `buttefly_circuit`), so the compiled reference has no symbol for them.  This translation unit includes that header FROM
// WHERE IT LIES (-I$(REF)/dilithium-256/hardware_code, oracle/Makefile `ref`), instantiates the two templates at the types every call
// site of the reference uses (<data2_t, data_t>: ntt2x2_fwdntt.cpp:102,152, ntt2x2_invntt.cpp:94, ntt2x2_mul.cpp:54) and gives the
// instantiations C names, so tests/test_oracle.py can pin the oracle's butterfly models against the reference's own unit.
// Nothing of the reference is restated here.
#include <stdint.h>

#include "config.h"            // enum OPERATION, as ntt2x2_mul.cpp:24-27 includes it ahead of the unit
#include "butterfly_unit.h"

extern "C" {

void ref_butterfly(int mode, int32_t* bj, int32_t* bjlen, int32_t zeta, int32_t aj, int32_t ajlen)
{
    data_t o0, o1;
    butterfly<data2_t, data_t>(static_cast<OPERATION>(mode), &o0, &o1, zeta, aj, ajlen);
    *bj = o0;
    *bjlen = o1;
}

void ref_buttefly_circuit(int32_t data_out[4], const int32_t data_in[4], const int32_t w[4], int mode)
{
    buttefly_circuit<data2_t, data_t>(data_out, data_in, w, static_cast<OPERATION>(mode));
}

}  // extern "C"
