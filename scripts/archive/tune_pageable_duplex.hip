// tune_pageable_duplex.hip -- can a PAGEABLE caller buffer use both directions of the host link at once?  The runtime stages pageable copies
// itself and blocks the calling thread while it does, so one thread alternating uploads and downloads gets the sum of the two (2.5 ms
// per 64 MiB + 64 MiB, profiles/r05g_host_pipe.txt).  Here: thread A uploads chunk after chunk, thread B downloads each chunk once its
// upload (and a stand-in kernel) is done -- the pattern a two-thread dil_ntt_host would use.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread scripts/tune_pageable_duplex.hip -o scripts/bin/tune_pageable_duplex
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void touch(int* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = 64u << 20;
    char* host = (char*)malloc(total);
    memset(host, 1, total);
    char* dev;
    CK(hipMalloc(&dev, total));
    hipStream_t up, dn;
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&dn, hipStreamNonBlocking));
    auto med = [&](auto f) {
        std::vector<double> t;
        for (int i = 0; i < 7; i++) {
            const double t0 = now();
            f();
            t.push_back(now() - t0);
        }
        std::sort(t.begin(), t.end());
        return t[3];
    };
    printf("64 MiB up + 64 MiB down, pageable (malloc) host buffer, in place\n");
    for (size_t chunk : {64u << 20, 16u << 20, 8u << 20, 4u << 20, 2u << 20, 1u << 20}) {
        const size_t nch = total / chunk;
        std::vector<hipEvent_t> ev(nch);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const double t_one = med([&] {          // one thread, one stream: up, kernel, down per chunk
            for (size_t c = 0; c < nch; c++) {
                CK(hipMemcpyAsync(dev + c * chunk, host + c * chunk, chunk, hipMemcpyHostToDevice, up));
                touch<<<1024, 256, 0, up>>>((int*)(dev + c * chunk), chunk / 4);
                CK(hipMemcpyAsync(host + c * chunk, dev + c * chunk, chunk, hipMemcpyDeviceToHost, up));
            }
            CK(hipStreamSynchronize(up));
        });
        const double t_two = med([&] {          // two threads
            std::atomic<size_t> uploaded{0};
            std::thread down([&] {
                for (size_t c = 0; c < nch; c++) {
                    while (uploaded.load(std::memory_order_acquire) <= c) std::this_thread::yield();
                    CK(hipStreamWaitEvent(dn, ev[c], 0));
                    CK(hipMemcpyAsync(host + c * chunk, dev + c * chunk, chunk, hipMemcpyDeviceToHost, dn));
                }
                CK(hipStreamSynchronize(dn));
            });
            for (size_t c = 0; c < nch; c++) {
                CK(hipMemcpyAsync(dev + c * chunk, host + c * chunk, chunk, hipMemcpyHostToDevice, up));
                touch<<<1024, 256, 0, up>>>((int*)(dev + c * chunk), chunk / 4);
                CK(hipEventRecord(ev[c], up));
                uploaded.store(c + 1, std::memory_order_release);
            }
            down.join();
            CK(hipStreamSynchronize(up));
        });
        printf("  chunk %6zu KiB: one thread %6.2f ms (%4.1f GB/s each way)   two threads %6.2f ms (%4.1f GB/s each way)\n", chunk >> 10, t_one * 1e3,
               total / t_one / 1e9, t_two * 1e3, total / t_two / 1e9);
        for (auto& e : ev) CK(hipEventDestroy(e));
    }
    return 0;
}
