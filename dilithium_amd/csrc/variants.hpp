// variants.hpp -- ablation and A/B switches of the kernels.  NOT part of the shipped build: device_common.hpp includes this
// file only under -DDIL_VARIANT_BUILD (scripts/build_variant.py adds it) and refuses any of the switches below without it,
// so that a stray -D can never put an ablation (inputs replaced by lane arithmetic, transforms removed) into libdil256.so.
//   DIL_ABL_NONTT     fused verify kernels without their forward / inverse transforms      (profiles/r02_fused_ab.txt 1)
//   DIL_ABL_NOALOAD   fused verify kernel with the matrix rows synthesised instead of loaded (same file: memory-only skeleton)
//   DIL_ABL_NOSMALL   no time-domain loads (z / c / t1 / y come from lane arithmetic)
//   DIL_ABL_A_PLAIN   default cache policy for the per-item matrix stream instead of non-temporal
// Results of the spent switches (row-major matrix, non-temporal w1 rows, strided-plain NTT accesses, the ExpandA / in-kernel
// ExpandA phase ablations) are in profiles/r02_fused_ab.txt, r02_expand_a.txt, r02_gen_a.txt; their code is gone.
#pragma once
#ifndef DIL_VARIANT_BUILD
#error "variants.hpp is for A/B builds only (scripts/build_variant.py)"
#endif

#ifdef DIL_ABL_NONTT
#define VW_FWD(r, tw, x) ((void)0)
#define VW_INV(r, tw, x) ((void)0)
#endif
#ifdef DIL_ABL_NOALOAD
#define VW_ALOAD(Ar, p, lane, st)                                                                            \
    do {                                                                                                     \
        for (int l_ = 0; l_ < L; l_++) Ar.v[l_] = make_int4(lane + l_, lane * 3, 7 * l_ + 1, lane ^ l_);     \
    } while (0)
#endif
#ifdef DIL_ABL_NOSMALL
#define DIL_LOAD_STRIDED_HOOK(r, a, lane)                       \
    do {                                                        \
        for (int m_ = 0; m_ < 4; m_++) r[m_] = lane * 17 + m_;  \
    } while (0)
#endif
#ifdef DIL_ABL_A_PLAIN
#define DIL_AROW_STREAM_HOOK(stream) (false)
#endif

// round 4: the shared-key mat-vec / sign phase 1 kernel taken apart (profiles/r04_mvs_ablation.txt)
//   DIL_ABL_MVS_NONTT   no forward / inverse transforms      DIL_ABL_MVS_NOINV / NOFWD: only one of them removed
//   DIL_ABL_MVS_NOAREAD the matrix operand from lane arithmetic instead of LDS
//   DIL_ABL_MVS_NOEMIT  the output stage replaced by one dword store per lane and row
#if defined(DIL_ABL_MVS_NONTT) || defined(DIL_ABL_MVS_NOFWD) || defined(DIL_ABL_MVS_NOINV)
#if defined(DIL_ABL_MVS_NONTT) || defined(DIL_ABL_MVS_NOFWD)
#define MVS_FWD(r, tw, x) ((void)0)
#define MVS_FWD2(a, b, tw, x) ((void)0)
#define MVS_FWDN(v, tw, x) ((void)0)
#else
#define MVS_FWD(r, tw, x) ntt_fwd_core(r, tw, x)
#define MVS_FWD2(a, b, tw, x) ntt_fwd_core2(a, b, tw, x)
#endif
#if defined(DIL_ABL_MVS_NONTT) || defined(DIL_ABL_MVS_NOINV)
#define MVS_INV(r, tw, x) ((void)0)
#define MVS_INV2(a, b, tw, x) ((void)0)
#else
#define MVS_INV(r, tw, x) ntt_inv_core(r, tw, x)
#define MVS_INV2(a, b, tw, x) ntt_inv_core2(a, b, tw, x)
#endif
#endif
#ifdef DIL_ABL_MVS_NOAREAD
#define MVS_AREAD(p) make_int4(lane + l, lane * 3 + k, 7 * l + 1, lane ^ k)
#endif
#ifdef DIL_ABL_MVS_NOEMIT
#define MVS_EMIT(call, w0, o, r, lane) st_nt(w0 + (o) + lane, r[0] ^ r[1] ^ r[2] ^ r[3])
#endif
