// copy_pool.hpp -- memcpy spread over the calling thread and a few parked pool threads (pure C++: the pool threads never call the HIP
// runtime).  The host-pointer entry points move a pageable caller buffer through the library's own page-locked slots with it (capi.hip
// "host-pointer entry points": the library never page-locks caller memory); one thread moves ~24 GB/s here, the link 57.
// (Round 6 measured a form in which pool threads and caller poll for ~200 us before they park: no gain on the GPU boxes -- configs[1] from
// pageable arrays 18.4 -> 19.1 M NTT/s, within noise -- and a loss under a tight CPU quota, where polling threads eat the copy's own cycles;
// the pool parks at once.)  Tested without a GPU under ThreadSanitizer: tests/cpp/test_copy_pool.cpp (tests/test_copy_pool.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace dil {
class CopyPool {
    struct Job { char* d; const char* s; size_t n; };
    std::mutex call_mu;                 // one parallel copy at a time (calls on different devices take turns)
    std::mutex mu;                      // guards jobs / next / quit
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    std::vector<Job> jobs;
    size_t next = 0;
    bool quit = false;
    std::atomic<size_t> pending{0};     // pieces of the current batch not yet copied

    void worker()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [this] { return quit || next < jobs.size(); });
            if (quit) return;
            const Job j = jobs[next++];
            lk.unlock();
            memcpy(j.d, j.s, j.n);
            const bool last = pending.fetch_sub(1, std::memory_order_acq_rel) == 1;
            lk.lock();
            if (last) cv_done.notify_all();
        }
    }

public:
    // `threads` = how many threads may take part, the calling one included (1 ... 8); a thread gets at least 512 KiB
    void copy(void* dst, const void* src, size_t n, int threads)
    {
        const int want = threads < 1 ? 1 : threads > 8 ? 8 : threads;
        const size_t parts = std::min<size_t>((size_t)want, n >> 19);
        if (parts <= 1) {
            memcpy(dst, src, n);
            return;
        }
        std::lock_guard<std::mutex> whole(call_mu);
        std::unique_lock<std::mutex> lk(mu);
        while (th.size() + 1 < parts) {
            try {
                th.emplace_back([this] { worker(); });
            } catch (const std::exception&) {
                break;                                       // fewer threads than asked for: the pieces get larger
            }
        }
        const size_t p = std::min(parts, th.size() + 1), piece = ((n / p) + 4095) & ~(size_t)4095;
        jobs.clear();
        next = 0;
        for (size_t i = 1; i < p; i++) {
            const size_t off = i * piece;
            if (off < n) jobs.push_back({static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(piece, n - off)});
        }
        pending.store(jobs.size(), std::memory_order_relaxed);
        cv_work.notify_all();
        lk.unlock();
        memcpy(dst, src, std::min(piece, n));
        lk.lock();
        cv_done.wait(lk, [this] { return pending.load(std::memory_order_acquire) == 0; });
    }
    size_t pool_threads()
    {
        std::lock_guard<std::mutex> lk(mu);
        return th.size();
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv_work.notify_all();
        }
        for (std::thread& t : th)
            if (t.joinable()) t.join();
    }
};
}  // namespace dil
