// membench.hip -- what streaming bandwidth does this MI355X actually deliver, by access shape?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { M_COPY = 0, M_INPLACE = 1, M_READ = 2, M_WRITE = 3 };

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n, uint32_t* sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint4 v = make_uint4(1, 2, 3, 4);
        if (MODE != M_WRITE) {
            if (NT) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(src + i);
                v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1);
                v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3);
            } else v = src[i];
        }
        if (MODE == M_READ) { acc += v.x ^ v.y ^ v.z ^ v.w; continue; }
        v.x += 1;
        if (NT) {
            uint32_t* p = reinterpret_cast<uint32_t*>(dst + i);
            __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1);
            __builtin_nontemporal_store(v.z, p + 2); __builtin_nontemporal_store(v.w, p + 3);
        } else dst[i] = v;
    }
    if (MODE == M_READ && acc == 0x12345678) *sink = acc;
}

// wave-chunked: each wave streams whole 1 KiB polynomials (the NTT kernels' shape), UNROLL in flight
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void kchunk(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nchunks)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    for (size_t c = wave * UNROLL; c < nchunks; c += nwaves * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (c + u < nchunks) v[u] = src[(c + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (c + u < nchunks) { v[u].x += 1; dst[(c + u) * 64 + lane] = v[u]; }
    }
}

template <class F>
float time_it(F&& launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch(i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = 1ull << 30;     // 1 GiB per buffer
    uint4 *a, *b;
    uint32_t* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const size_t n = bytes / 16;
    printf("%-52s %10s %12s\n", "variant (1 GiB buffers)", "us", "GB/s moved");
#define R(label, traffic, expr) { float ms = time_it([&](int i) { expr; }, 6); printf("%-52s %10.1f %12.1f\n", label, ms * 1e3, (traffic) / (ms * 1e-3) / 1e9); }
    for (int bpc : {2, 4, 8, 16, 32}) {
        char l[96];
        snprintf(l, 96, "copy a->b        bpc=%d", bpc);  R(l, 2.0 * bytes, (k16<M_COPY, false><<<cus * bpc, 256>>>(b, a, n, sink)));
        snprintf(l, 96, "copy a->b   nt   bpc=%d", bpc);  R(l, 2.0 * bytes, (k16<M_COPY, true><<<cus * bpc, 256>>>(b, a, n, sink)));
        snprintf(l, 96, "in-place         bpc=%d", bpc);  R(l, 2.0 * bytes, (k16<M_INPLACE, false><<<cus * bpc, 256>>>(a, a, n, sink)));
        snprintf(l, 96, "in-place    nt   bpc=%d", bpc);  R(l, 2.0 * bytes, (k16<M_INPLACE, true><<<cus * bpc, 256>>>(a, a, n, sink)));
        snprintf(l, 96, "read only        bpc=%d", bpc);  R(l, 1.0 * bytes, (k16<M_READ, false><<<cus * bpc, 256>>>(b, a, n, sink)));
        snprintf(l, 96, "read only   nt   bpc=%d", bpc);  R(l, 1.0 * bytes, (k16<M_READ, true><<<cus * bpc, 256>>>(b, a, n, sink)));
        snprintf(l, 96, "write only       bpc=%d", bpc);  R(l, 1.0 * bytes, (k16<M_WRITE, false><<<cus * bpc, 256>>>(b, a, n, sink)));
        snprintf(l, 96, "write only  nt   bpc=%d", bpc);  R(l, 1.0 * bytes, (k16<M_WRITE, true><<<cus * bpc, 256>>>(b, a, n, sink)));
    }
    for (int bpc : {4, 8}) {
        char l[96];
        snprintf(l, 96, "wave-chunk 1KiB x1 a->b      bpc=%d", bpc); R(l, 2.0 * bytes, (kchunk<1, false><<<cus * bpc, 256>>>(b, a, n / 64)));
        snprintf(l, 96, "wave-chunk 1KiB x2 a->b      bpc=%d", bpc); R(l, 2.0 * bytes, (kchunk<2, false><<<cus * bpc, 256>>>(b, a, n / 64)));
        snprintf(l, 96, "wave-chunk 1KiB x4 a->b      bpc=%d", bpc); R(l, 2.0 * bytes, (kchunk<4, false><<<cus * bpc, 256>>>(b, a, n / 64)));
        snprintf(l, 96, "wave-chunk 1KiB x4 in-place  bpc=%d", bpc); R(l, 2.0 * bytes, (kchunk<4, false><<<cus * bpc, 256>>>(a, a, n / 64)));
    }
    // hipMemcpy D2D as the vendor yardstick
    { float ms = time_it([&](int) { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 6);
      printf("%-52s %10.1f %12.1f\n", "hipMemcpyAsync D2D", ms * 1e3, 2.0 * bytes / (ms * 1e-3) / 1e9); }
    return 0;
}
