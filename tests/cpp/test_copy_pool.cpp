// tests/cpp/test_copy_pool.cpp -- dilithium_amd/csrc/copy_pool.hpp without a GPU (pure C++), meant to be built with -fsanitize=thread
// (tests/test_copy_pool.py): three caller threads issue copies of random sizes and thread counts against one pool, with pauses that let the
// pool threads park between batches; every destination is compared with its source.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "../../dilithium_amd/csrc/copy_pool.hpp"

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 60;
    dil::CopyPool pool;
    std::atomic<int> bad{0};
    std::atomic<long> copies{0};
    auto caller = [&](unsigned seed) {
        std::mt19937_64 rng(seed);
        std::vector<unsigned char> src(9u << 20), dst(9u << 20);
        for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned char)(rng() >> 7);
        for (int r = 0; r < rounds; r++) {
            const size_t n = (size_t)(rng() % (8u << 20)) + (rng() % 3 == 0 ? 0 : (1u << 19));
            const size_t so = rng() % 4096, doff = rng() % 4096;
            const int threads = 1 + (int)(rng() % 8);
            std::fill(dst.begin(), dst.end(), 0xA5);
            pool.copy(dst.data() + doff, src.data() + so, n, threads);
            if (memcmp(dst.data() + doff, src.data() + so, n) != 0) bad++;
            for (size_t i = 0; i < doff; i++)
                if (dst[i] != 0xA5) { bad++; break; }
            if (dst[doff + n] != 0xA5) bad++;                       // nothing past the end
            copies++;
            if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 1500));      // lets the pool park
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < 3; t++) th.emplace_back(caller, 1000 + t);
    for (auto& t : th) t.join();
    printf("copy pool: %ld copies from 3 caller threads, %zu pool threads, mismatches %d\n", copies.load(), pool.pool_threads(), bad.load());
    return bad ? 1 : 0;
}
