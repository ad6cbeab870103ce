// tune_keccak_coop.hip -- the wave-cooperative Keccak-f[1600] (csrc/keccak_coop.hpp) against the lane-per-sponge and two-lanes-per-sponge
// forms (csrc/keccak.hpp): correctness of the permutation against a host FIPS-202 model, then the time of a CHAIN of dependent
// permutations for N sponges, N = 1 ... 49152 -- the shape of the latency-bound kernels (H(mu || w1): 7-9 permutations, ExpandMask: 5).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/tune_keccak_coop.hip -o scripts/bin/tune_keccak_coop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../dilithium_amd/csrc/keccak.hpp"
#include "../dilithium_amd/csrc/keccak_coop.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// host model ---------------------------------------------------------------------------------------------------------
static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                                0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                                0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                                0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                                0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static uint64_t rol(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }
static void host_f1600(uint64_t* a)
{
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[r];
    }
}

// kernels: state[sponge][25] in, `perms` permutations, state out ---------------------------------------------------
__global__ __launch_bounds__(64) void coop_kernel(uint64_t* st, int perms, size_t n)
{
    const size_t sp = blockIdx.x;
    dil::coop::Lane k;
    k.init(threadIdx.x);
    uint32_t* s32 = reinterpret_cast<uint32_t*>(st + sp * 25);
    uint32_t v = k.dword >= 0 ? s32[k.dword] : 0u;
    for (int p = 0; p < perms; p++) v = dil::coop::permute(v, k);
    if (k.dword >= 0) s32[k.dword] = v;
}
__global__ __launch_bounds__(64) void lane_kernel(uint64_t* st, int perms, size_t n)
{
    const size_t sp = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (sp >= n) return;
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = st[sp * 25 + i];
    for (int p = 0; p < perms; p++) dil::keccak_f1600(a);
#pragma unroll
    for (int i = 0; i < 25; i++) st[sp * 25 + i] = a[i];
}
__global__ __launch_bounds__(64) void two_kernel(uint64_t* st, int perms, size_t n)
{
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x, sp = t >> 1;
    if (sp >= n) return;
    const bool hi = t & 1;
    uint32_t* s32 = reinterpret_cast<uint32_t*>(st + sp * 25);
    uint32_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = s32[2 * i + hi];
    for (int p = 0; p < perms; p++) dil::keccak2_f1600(a, hi);
#pragma unroll
    for (int i = 0; i < 25; i++) s32[2 * i + hi] = a[i];
}

template <class F>
static double time_us(F launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t NMAX = 49152;
    std::vector<uint64_t> h(NMAX * 25), ref;
    uint64_t s = 0x243F6A8885A308D3ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    for (int i = 0; i < 25; i++) h[i] = 0;                       // sponge 0: the all-zero state
    uint64_t* d;
    CK(hipMalloc(&d, NMAX * 25 * 8));
    // correctness: 1 and 3 permutations of 4096 states, all three forms
    const size_t NC = 4096;
    for (int perms : {1, 3}) {
        ref.assign(h.begin(), h.begin() + NC * 25);
        for (size_t i = 0; i < NC; i++)
            for (int p = 0; p < perms; p++) host_f1600(&ref[i * 25]);
        const char* names[3] = {"coop", "lane", "two-lane"};
        for (int form = 0; form < 3; form++) {
            CK(hipMemcpy(d, h.data(), NC * 25 * 8, hipMemcpyHostToDevice));
            if (form == 0) coop_kernel<<<NC, 64>>>(d, perms, NC);
            else if (form == 1) lane_kernel<<<(NC + 63) / 64, 64>>>(d, perms, NC);
            else two_kernel<<<(2 * NC + 63) / 64, 64>>>(d, perms, NC);
            CK(hipDeviceSynchronize());
            std::vector<uint64_t> got(NC * 25);
            CK(hipMemcpy(got.data(), d, NC * 25 * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < NC * 25; i++) bad += got[i] != ref[i];
            printf("check %-8s perms=%d: %s (%zu of %zu words differ)\n", names[form], perms, bad ? "MISMATCH" : "ok", bad, NC * 25);
            if (bad && form == 0) {
                for (int i = 0; i < 25; i++) printf("  w%02d got %016llx want %016llx\n", i, (unsigned long long)got[i], (unsigned long long)ref[i]);
            }
        }
    }
    // timing: a chain of 8 permutations (+ the state's load / store), N sponges
    const int perms = 8;
    printf("\nchain of %d permutations, median of 20 launches, us (and us per permutation)\n", perms);
    printf("%8s  %18s  %18s  %18s\n", "sponges", "coop (1/wave)", "two-lane (32/wave)", "lane (64/wave)");
    CK(hipMemcpy(d, h.data(), NMAX * 25 * 8, hipMemcpyHostToDevice));
    for (size_t n : {(size_t)1, (size_t)32, (size_t)64, (size_t)256, (size_t)1024, (size_t)2048, (size_t)2752, (size_t)4096, (size_t)5504, (size_t)8192, (size_t)16384, (size_t)24576, (size_t)49152}) {
        const double tc = time_us([&] { coop_kernel<<<n, 64>>>(d, perms, n); }, 20);
        const double t2 = time_us([&] { two_kernel<<<(2 * n + 63) / 64, 64>>>(d, perms, n); }, 20);
        const double t1 = time_us([&] { lane_kernel<<<(n + 63) / 64, 64>>>(d, perms, n); }, 20);
        printf("%8zu  %9.1f (%6.2f)  %9.1f (%6.2f)  %9.1f (%6.2f)\n", n, tc, tc / perms, t2, t2 / perms, t1, t1 / perms);
    }
    // the launch floor for scale
    const double t0 = time_us([&] { coop_kernel<<<1, 64>>>(d, 0, 1); }, 20);
    printf("empty launch (0 permutations): %.1f us\n", t0);
    return 0;
}
