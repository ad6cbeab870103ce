// capi_internal.hpp -- what the translation units behind include/dil256.h share: the per-process runtime state,
// lazy initialisation, error propagation.
#pragma once
#include "../../include/dil256.h"
#include "kernels.hpp"

#include <mutex>

namespace dil {
namespace rt {

struct State {
    std::mutex mu;
    bool ready = false;
    int device = -1;
    uint32_t* d_tables = nullptr;   // fwd | inv | inv_pipe
    dil::Tables t;
    void* scratch = nullptr;        // for *_host entry points
    size_t scratch_bytes = 0;
    int sign_early = 1;             // DIL_SIGN_EARLY: 0 = the signing loop evaluates every check of every attempt
    int sign_waste = 6144;          // DIL_SIGN_WASTE: speculative entries a round may expect to waste (see dil_sign_dev)
    int sign_cap = 0;               // DIL_SIGN_CAP: entries in flight per signing round (0 = default)
    int aux_overlap = 1;            // DIL_AUX_OVERLAP: 0 = composite calls never use the helper stream
    int sign_streams = 1;           // DIL_SIGN_STREAMS: 2 = split each signing round over the caller stream and a helper (measured: no gain)
};
extern State g;

int ensure_init();
void release_scratch();        // frees the composite calls' per-stream arenas (scheme.hip)
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace rt
}  // namespace dil

#define DIL_TRY(expr)                          \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)
