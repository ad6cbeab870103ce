// launch_util.hpp -- host-side grid sizing shared by the launchers
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <atomic>
#include <mutex>
#include <stdint.h>
#include <string.h>
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

// Debug record of the most recent launch of every PERSISTENT kernel family (grid = min(work, resident)): the parity
// tests read it through dil_launch_info() to assert that a batch was large enough for the kernels' item loops to be
// re-entered -- the code after the first item (next-item prefetch, loop-carried registers) is otherwise never compared
// with the oracle.  Three relaxed atomic stores per launch; process-wide (the tests are single-threaded).
struct LaunchRecord {
    const char* family;
    std::atomic<uint32_t> grid{0}, items_per_block{0};
    std::atomic<uint64_t> items{0}, launches{0};
};
inline LaunchRecord* launch_records(int* count)
{
    static LaunchRecord tab[] = {{"matvec_wpi"},  {"sign1_wpi"},     {"matvec_shared"},   {"sign1_shared"},       {"keygen_wpi"},
                                 {"verify_wpi"},  {"verify_shared"}, {"sign2_wpi"},       {"sign2_early_wpi"},    {"verify_wire_wpi"},
                                 {"verify_wire_shared"}};
    if (count) *count = (int)(sizeof(tab) / sizeof(tab[0]));
    return tab;
}
inline void note_launch(const char* family, int grid, int items_per_block, size_t items)
{
    int n;
    LaunchRecord* tab = launch_records(&n);
    for (int i = 0; i < n; i++)
        if (!strcmp(tab[i].family, family)) {
            tab[i].grid.store((uint32_t)grid, std::memory_order_relaxed);
            tab[i].items_per_block.store((uint32_t)items_per_block, std::memory_order_relaxed);
            tab[i].items.store(items, std::memory_order_relaxed);
            tab[i].launches.fetch_add(1, std::memory_order_relaxed);
            return;
        }
}

// (A "balanced" grid -- the largest one that gives every block the same number of passes, 683 instead of 768 workgroups for 8192
//  level-3 verifications -- was measured slower, 63.5 vs 58.5 us: more waves in flight beat an even tail.)
static inline int grid_for(size_t work_blocks, int max_blocks)
{
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks < (size_t)max_blocks ? work_blocks : (size_t)max_blocks);
}

}  // namespace dil
