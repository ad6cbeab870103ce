// device_common.hpp -- small device helpers shared by kernels.hip and pipelines.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Ablation / A-B switches live in variants.hpp, which only a variant build sees (scripts/build_variant.py); the shipped
// build refuses them outright.
#ifdef DIL_VARIANT_BUILD
#include "variants.hpp"
#elif defined(DIL_ABL_NONTT) || defined(DIL_ABL_NOALOAD) || defined(DIL_ABL_NOSMALL) || defined(DIL_ABL_A_PLAIN) || \
    defined(DIL_ABL_AROWMAJOR) || defined(DIL_ABL_W1_NT) || defined(DIL_NTT_STRIDED_PLAIN) || defined(DIL_EA_ABL) || defined(DIL_GEN_ABL)
#error "ablation switches need a variant build: scripts/build_variant.py <name> -DDIL_ABL_... (adds -DDIL_VARIANT_BUILD)"
#endif

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
