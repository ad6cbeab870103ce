// multi_gpu.hip -- the C++ multi-GPU host layer (north_star "host code in C++"): independent polynomials /
// signatures shard embarrassingly, so a host batch is cut into contiguous slices [g*B/G, (g+1)*B/G) (sizes differ by at
// most one), one host thread per device runs the single-device host-pointer entry point on its slice with that device
// current -- per-device runtime state (capi_internal.hpp) makes the threads independent -- and the "gather" is each
// thread's D2H copy into the caller's array.  No collective: within one process the slabs meet in host memory; across
// processes (one per GPU, torch.distributed / RCCL) dilithium_amd/sharding.py does the same slicing and all-gathers.
#include "../../include/dil256.h"

#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

extern "C" void dil_shard_range(size_t n_items, int rank, int world, size_t* lo, size_t* hi)
{
    if (world < 1) world = 1;
    const size_t base = n_items / (size_t)world, rem = n_items % (size_t)world;
    const size_t r = (size_t)rank;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

namespace {
int device_count(int ndev)
{
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return -1;
    return (ndev <= 0 || ndev > have) ? have : ndev;
}

// run fn(lo, hi) for every device's slice on its own thread with that device current; first error wins
template <class F>
int for_each_device(size_t batch, int ndev, F&& fn)
{
    const int G = device_count(ndev);
    if (G < 0) return (int)hipErrorNoDevice;
    if (batch == 0) return 0;
    std::vector<int> rc((size_t)G, 0);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        th.emplace_back([&, g] {
            size_t lo, hi;
            dil_shard_range(batch, g, G, &lo, &hi);
            if (lo == hi) return;
            const hipError_t e = hipSetDevice(g);
            rc[(size_t)g] = e != hipSuccess ? (int)e : fn(lo, hi);
        });
    }
    for (std::thread& t : th) t.join();
    for (int r : rc)
        if (r) return r;
    return 0;
}
}  // namespace

extern "C" {

int dil_ntt_multi_host(int32_t* polys, size_t batch, int inverse, int ndev)
{
    return for_each_device(batch, ndev, [&](size_t lo, size_t hi) {
        return inverse ? dil_invntt_host(polys + lo * 256, hi - lo) : dil_ntt_host(polys + lo * 256, hi - lo);
    });
}

int dil_keygen_multi_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, int ndev)
{
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, ndev, [&](size_t lo, size_t hi) {
        return dil_keygen_host(pk + lo * pkb, sk + lo * skb, seed + lo * 32, level, hi - lo);
    });
}

int dil_sign_multi_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                        int max_attempts, int ndev)
{
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    if (!skb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, ndev, [&](size_t lo, size_t hi) {
        return dil_sign_host(sig + lo * sgb, attempts ? attempts + lo : nullptr, shared_sk ? sk : sk + lo * skb, mu + lo * 64, level,
                             hi - lo, shared_sk, max_attempts);
    });
}

int dil_verify_sig_multi_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                              int shared_pk, int ndev)
{
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, ndev, [&](size_t lo, size_t hi) {
        return dil_verify_sig_host(verdict + lo, shared_pk ? pk : pk + lo * pkb, sig + lo * sgb, mu + lo * 64, level, hi - lo, shared_pk);
    });
}

}  // extern "C"
