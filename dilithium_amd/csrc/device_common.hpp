// device_common.hpp -- small device helpers shared by kernels.hip and pipelines.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (The ablation / A-B switches of rounds 2-5 -- variants.hpp, scripts/build_variant.py -- are resolved to the shipped arm: their experiments are
//  closed, the results are in profiles/r02_fused_ab.txt, r04_mvs_ablation.txt, r05j_ab_verify_sets.txt.)

namespace dil {

// Streaming accesses use the non-temporal cache policy: every polynomial is touched exactly
// once per kernel, and measured on MI355X nt loads + stores lift the in-place 1 KiB-in /
// 1 KiB-out stream from 4.7 to 5.2 TB/s (profiles/r01_tune_ntt.txt).
__device__ __forceinline__ int32_t ld_nt(const int32_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(int32_t* p, int32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ int4 ld_nt4(const int32_t* p)
{
    int4 v;
    v.x = __builtin_nontemporal_load(p);
    v.y = __builtin_nontemporal_load(p + 1);
    v.z = __builtin_nontemporal_load(p + 2);
    v.w = __builtin_nontemporal_load(p + 3);
    return v;      // hipcc merges the four into one global_load_dwordx4 ... nt
}
__device__ __forceinline__ void st_nt4(int32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    __builtin_nontemporal_store((int32_t)a, p);
    __builtin_nontemporal_store((int32_t)b, p + 1);
    __builtin_nontemporal_store((int32_t)c, p + 2);
    __builtin_nontemporal_store((int32_t)d, p + 3);
}

}  // namespace dil
