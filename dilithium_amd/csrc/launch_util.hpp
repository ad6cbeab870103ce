// launch_util.hpp -- host-side grid sizing shared by the launchers
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <mutex>
#include <vector>

namespace dil {

// Resident blocks per CU of a kernel (occupancy API): persistent grids are sized to what is actually co-resident so
// that no block waits for another to retire.  Cached per (device, kernel, block size) in one mutex-protected table
// shared by every translation unit -- a lookup is a few comparisons, the occupancy query runs once per key.
struct OccKey {
    int device;
    const void* kernel;
    int threads;
    int blocks;
};
inline int occ_cache(int device, const void* kernel, int threads, int set_blocks)   // set_blocks < 0: look up (-1 = miss)
{
    static std::mutex mu;
    static std::vector<OccKey> tab;
    std::lock_guard<std::mutex> lk(mu);
    for (const OccKey& k : tab)
        if (k.kernel == kernel && k.device == device && k.threads == threads) return k.blocks;
    if (set_blocks < 0) return -1;
    tab.push_back(OccKey{device, kernel, threads, set_blocks});
    return set_blocks;
}

template <class KernelT>
inline int resident_blocks_per_cu(KernelT kernel, int block_threads, int cap, int device)
{
    const void* key = reinterpret_cast<const void*>(kernel);
    int n = occ_cache(device, key, block_threads, -1);
    if (n < 0) {
        n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block_threads, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        n = occ_cache(device, key, block_threads, n);
    }
    return n < cap ? n : cap;
}

static inline int grid_for(size_t work_blocks, int max_blocks)
{
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks < (size_t)max_blocks ? work_blocks : (size_t)max_blocks);
}

}  // namespace dil
