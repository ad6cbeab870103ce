// tune_vgpr_bank.hip -- does the issue rate of a lone wavefront's three-source VALU instructions (v_bitop3_b32, v_alignbit_b32: 2/3 of a
// Keccak round) depend on WHICH registers the sources are?  The lane-per-sponge Keccak runs at 4.9 cycles per instruction where the
// issue limit is 4; if same-bank sources cost a cycle, register allocation is a lever.   One wave, 64 instructions per loop trip,
// sources chosen by the pattern, destinations rotating over 8 registers nobody reads (no dependence chains).
//   hipcc --offload-arch=gfx950 -O2 scripts/tune_vgpr_bank.hip -o scripts/bin/tune_vgpr_bank
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define I8(OP, A, B, C, X)                                                                                                        \
    OP " v40, " A ", " B ", " C X "\n" OP " v41, " A ", " B ", " C X "\n" OP " v42, " A ", " B ", " C X "\n" OP " v43, " A ", " B ", " C X "\n" \
    OP " v44, " A ", " B ", " C X "\n" OP " v45, " A ", " B ", " C X "\n" OP " v46, " A ", " B ", " C X "\n" OP " v47, " A ", " B ", " C X "\n"
#define I64(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X) I8(OP, A, B, C, X)
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v20", "v24"

// dependent chain variants: each instruction reads the previous one's destination
#define D8(OP, B, C, X)                                                                                                            \
    OP " v40, v47, " B ", " C X "\n" OP " v41, v40, " B ", " C X "\n" OP " v42, v41, " B ", " C X "\n" OP " v43, v42, " B ", " C X "\n" \
    OP " v44, v43, " B ", " C X "\n" OP " v45, v44, " B ", " C X "\n" OP " v46, v45, " B ", " C X "\n" OP " v47, v46, " B ", " C X "\n"
#define D64(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X) D8(OP, B, C, X)

template <int V>
__global__ __launch_bounds__(1024) void probe(uint64_t* out, int trips)
{
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; i++) {
        if (V == 0) asm volatile(I64("v_bitop3_b32", "v8", "v9", "v10", " bitop3:0x96") ::: CLOB);        // banks 0 1 2
        if (V == 1) asm volatile(I64("v_bitop3_b32", "v8", "v12", "v16", " bitop3:0x96") ::: CLOB);       // banks 0 0 0
        if (V == 2) asm volatile(I64("v_bitop3_b32", "v8", "v12", "v9", " bitop3:0x96") ::: CLOB);        // banks 0 0 1
        if (V == 3) asm volatile(I64("v_bitop3_b32", "v8", "v8", "v8", " bitop3:0x96") ::: CLOB);         // one register
        if (V == 4) asm volatile(I64("v_alignbit_b32", "v8", "v9", "v10", "") ::: CLOB);
        if (V == 5) asm volatile(I64("v_alignbit_b32", "v8", "v12", "v16", "") ::: CLOB);
        if (V == 6) asm volatile(I64("v_alignbit_b32", "v8", "v9", "7", "") ::: CLOB);                    // constant shift
        if (V == 7) asm volatile(I64("v_alignbit_b32", "v8", "v12", "7", "") ::: CLOB);
        if (V == 8) asm volatile(I64("v_xor_b32", "v8", "v9", "", "") ::: CLOB);                          // (C empty: two sources)
        if (V == 9) asm volatile(I64("v_xor_b32", "v8", "v12", "", "") ::: CLOB);
        if (V == 10) asm volatile(D64("v_bitop3_b32", "v9", "v10", " bitop3:0x96") ::: CLOB);             // dependent chain
        if (V == 11) asm volatile(D64("v_alignbit_b32", "v9", "7", "") ::: CLOB);
        if (V == 12) asm volatile(I64("v_bitop3_b32", "v40", "v9", "v10", " bitop3:0x96") ::: CLOB);      // a source that is also a (rotating) destination
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

int main()
{
    uint64_t* d;
    CK(hipMalloc(&d, 8));
    const int trips = 2000;
    const char* names[] = {"v_bitop3_b32   sources in banks 0 1 2", "v_bitop3_b32   sources in banks 0 0 0", "v_bitop3_b32   sources in banks 0 0 1",
                           "v_bitop3_b32   one register three times", "v_alignbit_b32 sources in banks 0 1 2", "v_alignbit_b32 sources in banks 0 0 0",
                           "v_alignbit_b32 banks 0 1, constant shift", "v_alignbit_b32 banks 0 0, constant shift", "v_xor_b32      banks 0 1",
                           "v_xor_b32      banks 0 0", "v_bitop3_b32   dependent chain", "v_alignbit_b32 dependent chain", "v_bitop3_b32   one source = a recent destination"};
    // s_memtime / readcyclecounter counts at a constant 100 MHz on this part: use wall time instead (events) and the shader clock of the guide
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(V)                                                                                                            \
    {                                                                                                                     \
        printf("%-48s", names[V]);                                                                                        \
        for (int threads : {64, 256, 512, 1024}) {          /* 1 wave; 1, 2, 4 waves on every SIMD of one CU */           \
            probe<V><<<1, threads>>>(d, 200);                                                                             \
            CK(hipDeviceSynchronize());                                                                                   \
            CK(hipEventRecord(e0));                                                                                       \
            probe<V><<<1, threads>>>(d, trips * 10);                                                                      \
            CK(hipEventRecord(e1));                                                                                       \
            CK(hipEventSynchronize(e1));                                                                                  \
            float ms;                                                                                                     \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                                         \
            const double ns_per = (ms * 1e6 - 8000.0) / (trips * 10.0 * 64.0);                                            \
            printf("  %6.3f", ns_per);                                                                                    \
        }                                                                                                                 \
        printf("\n");                                                                                                     \
    }
    printf("ns per instruction of ONE wave, with 1 wave in the CU | 1 | 2 | 4 waves on each of its SIMDs\n");
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    return 0;
}
