// launch_util.hpp -- host-side grid sizing shared by the launchers
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dil {

// resident blocks per CU of a kernel (occupancy API, cached per kernel): persistent grids are
// sized to what is actually co-resident so that no block waits for another to retire
template <class KernelT>
static inline int resident_blocks_per_cu(KernelT kernel, int block_threads, int cap)
{
    static int cached = 0;          // one instance per KernelT instantiation... but KernelT is a type:
    static const void* cached_for = nullptr;
    const void* key = reinterpret_cast<const void*>(kernel);
    if (cached_for != key) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block_threads, 0) != hipSuccess || n < 1) n = 1;
        cached = n;
        cached_for = key;
    }
    return cached < cap ? cached : cap;
}

static inline int grid_for(size_t work_blocks, int max_blocks)
{
    if (work_blocks < 1) work_blocks = 1;
    return (int)(work_blocks < (size_t)max_blocks ? work_blocks : (size_t)max_blocks);
}


}  // namespace dil
