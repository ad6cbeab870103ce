// coop_kernels.hip -- the SHAKE-bound kernels in their one-sponge-per-wavefront form (keccak_coop.hpp, coop_bodies.hpp): what the launchers
// of hash_kernels.hip / wire_kernels.hip / codec_kernels.hip run while a call has few sponges (option coop_max; see coop_wanted()).
// One workgroup = one wave = one sponge.  Parity: tests/test_gpu_coop.py runs every entry point with the option forced both ways
// (cooperative at every size / never) at 1 ... 49152 sponges against each other, hashlib and the host samplers; tests/test_gpu_options.py
// re-runs the scheme-level KAT tests with coop_max = 0 and 2^30.
#include "coop_bodies.hpp"
#include "kernels.hpp"

namespace dil {

// A call with at most this many sponges runs them one per wavefront.  Crossover against the two-lane form for a chain of eight
// permutations: 2752 sponges 43 vs 54 us, 4096 53 vs 54 (profiles/r05d_keccak_coop_asm.txt).  0: never.
std::atomic<int> coop_max_sponges{3072};
bool coop_wanted(size_t sponges) { return sponges > 0 && sponges <= (size_t)coop_max_sponges.load(std::memory_order_relaxed); }
// SampleInBall is ONE permutation and then a sampler that the cooperative form runs in parallel (ballots + a lane per sign) where the
// lane-per-item form walks 60-odd bytes serially: alone it wins up to ~10000 items (4096: 12.9 vs 24.8 us, 8192: 21 vs 25, 16384: 37 vs
// 26; scripts/bench_sib_forms.py) -- but it is eight times the issue slots, and beside a throughput-bound neighbour (ExpandA of 8192 keys
// on the other stream) the call gets SLOWER from 8192 items on (281.7 vs 274.6 us): the bound is 1.5 x coop_max.
bool coop_wanted_sib(size_t items) { return items > 0 && 2 * items <= 3 * (size_t)coop_max_sponges.load(std::memory_order_relaxed); }

__global__ __launch_bounds__(64) void coop_shake256_kernel(uint32_t* __restrict__ out, int out_words, const uint32_t* __restrict__ in, int in_words)
{
    const size_t i = blockIdx.x;
    coop::shake256_body(out + i * (size_t)out_words * 2, out_words, in + i * (size_t)in_words * 2, in_words);
}

__global__ __launch_bounds__(64) void coop_challenge_hash_kernel(uint32_t* __restrict__ out32, int32_t* __restrict__ verdict, const uint32_t* __restrict__ mu,
                                                                 const uint32_t* __restrict__ w1p, int w1_words, const uint8_t* __restrict__ expect,
                                                                 size_t expect_stride)
{
    coop::challenge_hash_body(out32, verdict, mu, w1p, w1_words, expect, expect_stride, blockIdx.x);
}

__global__ __launch_bounds__(64) void coop_challenge_sample_kernel(uint32_t* __restrict__ ctilde_out, int32_t* __restrict__ c_out,
                             const uint32_t* __restrict__ mu,
                                                                   const uint32_t* __restrict__ w1p, int w1_words, int tau)
{
    __shared__ __attribute__((aligned(16))) coop::SibShared sh;
    coop::challenge_sample_body(ctilde_out, c_out, mu, w1p, w1_words, tau, blockIdx.x, sh);
}

template <bool BITS>      // BITS: c in the compact per-lane form of the wire-format verify kernels, else a canonical int32 polynomial
__global__ __launch_bounds__(64) void coop_sample_in_ball_kernel(void* __restrict__ out, const uint8_t* __restrict__ ctilde, size_t ct_stride, int tau)
{
    __shared__ __attribute__((aligned(16))) coop::SibShared sh;
    const size_t item = blockIdx.x;
    coop::Sponge<17> sp;
    sp.init(threadIdx.x);
    coop::sib_seed(sp, ctilde + item * ct_stride);
    coop::sib_sample(sp, tau, sh, threadIdx.x);
    if (BITS) coop::sib_store_bits(static_cast<uint32_t*>(out) + item * 64, sh, threadIdx.x);
    else coop::sib_store_poly(static_cast<int32_t*>(out) + item * 256, sh, threadIdx.x);
}

template <int B, bool RAW>
__global__ __launch_bounds__(64) void coop_expand_mask_kernel(void* __restrict__ y, const uint32_t* __restrict__ rhoprime,
                             const uint32_t* __restrict__ kappa, int L)
{
    __shared__ uint32_t stream[8 * B + 2];
    const size_t p = blockIdx.x, item = p / (size_t)L;
    const uint32_t nonce = (kappa[item] + (uint32_t)(p % (size_t)L)) & 0xFFFFu;
    if (RAW) coop::expand_mask_raw_body<B>(static_cast<uint32_t*>(y) + p * (8 * B), rhoprime + item * 16, nonce);
    else coop::expand_mask_body<B>(static_cast<int32_t*>(y) + p * 256, rhoprime + item * 16, nonce, stream);
}

__global__ __launch_bounds__(64) void coop_expand_a_kernel(int32_t* __restrict__ A, const uint32_t* __restrict__ rho, size_t rho_stride_dwords, int K, int L)
{
    __shared__ uint32_t blk[44];
    const size_t p = blockIdx.x, item = p / (size_t)(K * L);
    const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
    coop::expand_a_body(A + p * 256, rho + item * rho_stride_dwords, (uint32_t)j | ((uint32_t)i << 8), blk);
}

template <int ETA>
__global__ __launch_bounds__(64) void coop_expand_s_kernel(int32_t* __restrict__ s1, int32_t* __restrict__ s2, int L, int K,
                             const uint8_t* __restrict__ rhoprime,
                                                           size_t rp_stride)
{
    __shared__ uint32_t blk[36];
    const size_t p = blockIdx.x, item = p / (size_t)(L + K);
    const int j = (int)(p % (size_t)(L + K));
    int32_t* out = j < L ? s1 + (item * L + j) * 256 : s2 + (item * K + (j - L)) * 256;
    coop::expand_s_body<ETA>(out, rhoprime + item * rp_stride, (uint32_t)j, blk);
}

__global__ __launch_bounds__(64) void coop_mu_kernel(uint32_t* __restrict__ mu, const uint8_t* __restrict__ tr, size_t tr_stride,
                             const uint8_t* __restrict__ msgs,
                                                     size_t msgs_bytes, const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ lengths,
                                                     int32_t* __restrict__ bad)
{
    const size_t i = blockIdx.x;
    // an item whose (offset, length) leaves the blob is never read: it is hashed as an EMPTY message and flagged
    const uint64_t off = offsets[i];
    const uint32_t len = lengths[i];
    const bool inside = off <= msgs_bytes && len <= msgs_bytes - off;
    if (bad && threadIdx.x == 0) bad[i] = inside ? 0 : 1;
    coop::mu_body(mu + i * 16, reinterpret_cast<const uint32_t*>(tr + i * tr_stride), inside ? msgs + off : msgs, inside ? len : 0u);
}

// Composite launches of the few-key paths, as in wire_kernels.hip / hash_kernels.hip: independent latency-bound jobs side by side in ONE launch.
//   verification: A = ExpandA(rho) of the key(s) beside c = SampleInBall(c~) of the signatures
__global__ __launch_bounds__(64) void coop_expand_a_sib_kernel(int32_t* __restrict__ A, const uint32_t* __restrict__ rho, size_t rho_stride_dwords,
                             int K, int L,
                                                               unsigned a_blocks, uint32_t* __restrict__ cbits, const uint8_t* __restrict__ ctilde,
                                                               size_t ct_stride, int tau)
{
    __shared__ __attribute__((aligned(16))) coop::SibShared sh;
    if (blockIdx.x < a_blocks) {
        const size_t p = blockIdx.x, item = p / (size_t)(K * L);
        const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
        coop::expand_a_body(A + p * 256, rho + item * rho_stride_dwords, (uint32_t)j | ((uint32_t)i << 8), reinterpret_cast<uint32_t*>(sh.c));
        return;
    }
    const size_t item = blockIdx.x - a_blocks;
    coop::Sponge<17> sp;
    sp.init(threadIdx.x);
    coop::sib_seed(sp, ctilde + item * ct_stride);
    coop::sib_sample(sp, tau, sh, threadIdx.x);
    coop::sib_store_bits(cbits + item * 64, sh, threadIdx.x);
}
//   key generation: A = ExpandA(rho) beside (s1, s2) = ExpandS(rho')
template <int ETA>
__global__ __launch_bounds__(64) void coop_expand_a_s_kernel(int32_t* __restrict__ A, const uint32_t* __restrict__ rho, size_t rho_stride_dwords, int K, int L,
                                                             unsigned a_blocks, int32_t* __restrict__ s1, int32_t* __restrict__ s2,
                                                             const uint8_t* __restrict__ rhoprime, size_t rp_stride)
{
    __shared__ uint32_t blk[44];
    if (blockIdx.x < a_blocks) {
        const size_t p = blockIdx.x, item = p / (size_t)(K * L);
        const int ij = (int)(p % (size_t)(K * L)), i = ij / L, j = ij % L;
        coop::expand_a_body(A + p * 256, rho + item * rho_stride_dwords, (uint32_t)j | ((uint32_t)i << 8), blk);
        return;
    }
    const size_t p = blockIdx.x - a_blocks, item = p / (size_t)(L + K);
    const int j = (int)(p % (size_t)(L + K));
    int32_t* out = j < L ? s1 + (item * L + j) * 256 : s2 + (item * K + (j - L)) * 256;
    coop::expand_s_body<ETA>(out, rhoprime + item * rp_stride, (uint32_t)j, blk);
}

// ---- launchers (the callers have validated level / alignment) ----------------------------------------------------------------
static inline bool grid_ok(size_t n) { return n > 0 && n <= 0x7fffffffull; }

hipError_t launch_coop_shake256(uint64_t* out, int out_bytes, const uint64_t* in, int in_bytes, size_t batch, hipStream_t s)
{
    if (!grid_ok(batch)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_shake256_kernel, (unsigned)batch, 64, 0, s, reinterpret_cast<uint32_t*>(out), out_bytes / 8, reinterpret_cast<const uint32_t*>(in),
                       in_bytes / 8);
    return hipGetLastError();
}
hipError_t launch_coop_challenge_hash(uint8_t* out32, int32_t* verdict, const uint8_t* mu, const uint8_t* w1p, int w1_words, const uint8_t* expect,
                                      size_t expect_stride, size_t batch, hipStream_t s)
{
    if (!grid_ok(batch)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_challenge_hash_kernel, (unsigned)batch, 64, 0, s, reinterpret_cast<uint32_t*>(out32), verdict,
                       reinterpret_cast<const uint32_t*>(mu),
                       reinterpret_cast<const uint32_t*>(w1p), w1_words, expect, expect_stride);
    return hipGetLastError();
}
hipError_t launch_coop_challenge_sample(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1p, int w1_words, int tau, size_t batch, hipStream_t s)
{
    if (!grid_ok(batch)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_challenge_sample_kernel, (unsigned)batch, 64, 0, s, reinterpret_cast<uint32_t*>(ctilde), c, reinterpret_cast<const uint32_t*>(mu),
                       reinterpret_cast<const uint32_t*>(w1p), w1_words, tau);
    return hipGetLastError();
}
hipError_t launch_coop_sample_in_ball(int32_t* c, uint32_t* cbits, const uint8_t* ctilde, size_t ct_stride, int tau, size_t nitems, hipStream_t s)
{
    if (!grid_ok(nitems)) return hipErrorInvalidValue;
    if (cbits) hipLaunchKernelGGL(coop_sample_in_ball_kernel<true>, (unsigned)nitems, 64, 0, s, static_cast<void*>(cbits), ctilde, ct_stride, tau);
    else hipLaunchKernelGGL(coop_sample_in_ball_kernel<false>, (unsigned)nitems, 64, 0, s, static_cast<void*>(c), ctilde, ct_stride, tau);
    return hipGetLastError();
}
hipError_t launch_coop_expand_mask(void* y, bool raw, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s)
{
    const int L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const size_t total = nitems * (size_t)L;
    if (!grid_ok(total)) return hipErrorInvalidValue;
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(rhoprime);
    if (level == 2) {
        if (raw) hipLaunchKernelGGL((coop_expand_mask_kernel<18, true>), (unsigned)total, 64, 0, s, y, rp, kappa, L);
        else hipLaunchKernelGGL((coop_expand_mask_kernel<18, false>), (unsigned)total, 64, 0, s, y, rp, kappa, L);
    } else {
        if (raw) hipLaunchKernelGGL((coop_expand_mask_kernel<20, true>), (unsigned)total, 64, 0, s, y, rp, kappa, L);
        else hipLaunchKernelGGL((coop_expand_mask_kernel<20, false>), (unsigned)total, 64, 0, s, y, rp, kappa, L);
    }
    return hipGetLastError();
}
hipError_t launch_coop_expand_a(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int K, int L, size_t nitems, hipStream_t s)
{
    const size_t total = nitems * (size_t)(K * L);
    if (!grid_ok(total)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_expand_a_kernel, (unsigned)total, 64, 0, s, A, reinterpret_cast<const uint32_t*>(rho), rho_stride_bytes / 4, K, L);
    return hipGetLastError();
}
hipError_t launch_coop_expand_s(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride, int eta, int L, int K, size_t nitems, hipStream_t s)
{
    const size_t total = nitems * (size_t)(L + K);
    if (!grid_ok(total)) return hipErrorInvalidValue;
    if (eta == 2) hipLaunchKernelGGL(coop_expand_s_kernel<2>, (unsigned)total, 64, 0, s, s1, s2, L, K, rhoprime, rp_stride);
    else hipLaunchKernelGGL(coop_expand_s_kernel<4>, (unsigned)total, 64, 0, s, s1, s2, L, K, rhoprime, rp_stride);
    return hipGetLastError();
}
hipError_t launch_coop_expand_a_sib(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, size_t nkeys, int K, int L, uint32_t* cbits, const uint8_t* ctilde,
                                    size_t ct_stride, int tau, size_t nitems, hipStream_t s)
{
    const size_t a_blocks = nkeys * (size_t)(K * L);
    if (!grid_ok(a_blocks + nitems)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_expand_a_sib_kernel, (unsigned)(a_blocks + nitems), 64, 0, s, A, reinterpret_cast<const uint32_t*>(rho), rho_stride_bytes / 4, K, L,
                       (unsigned)a_blocks, cbits, ctilde, ct_stride, tau);
    return hipGetLastError();
}
hipError_t launch_coop_expand_a_s(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride,
                                  int K, int L, int eta, size_t nkeys, hipStream_t s)
{
    const size_t a_blocks = nkeys * (size_t)(K * L), s_blocks = nkeys * (size_t)(L + K);
    if (!grid_ok(a_blocks + s_blocks)) return hipErrorInvalidValue;
    if (eta == 2)
        hipLaunchKernelGGL(coop_expand_a_s_kernel<2>, (unsigned)(a_blocks + s_blocks), 64, 0, s, A, reinterpret_cast<const uint32_t*>(rho),
                           rho_stride_bytes / 4, K, L,
                           (unsigned)a_blocks, s1, s2, rhoprime, rp_stride);
    else
        hipLaunchKernelGGL(coop_expand_a_s_kernel<4>, (unsigned)(a_blocks + s_blocks), 64, 0, s, A, reinterpret_cast<const uint32_t*>(rho),
                           rho_stride_bytes / 4, K, L,
                           (unsigned)a_blocks, s1, s2, rhoprime, rp_stride);
    return hipGetLastError();
}
hipError_t launch_coop_mu(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                          const uint32_t* lengths, int32_t* bad, size_t batch, hipStream_t s)
{
    if (!grid_ok(batch)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coop_mu_kernel, (unsigned)batch, 64, 0, s, reinterpret_cast<uint32_t*>(mu), tr, tr_stride, msgs, msgs_bytes, offsets, lengths, bad);
    return hipGetLastError();
}

}  // namespace dil
