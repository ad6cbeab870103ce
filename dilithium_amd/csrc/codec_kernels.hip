// codec_kernels.hip -- rows N2 and N4 of SURVEY 8(f): the reference's bit-packing codecs and the
// keygen-side sampler on the device, in the wire formats its KAT files use (round-3 v3.1):
//   decoder.v:89-143 / uncenter_coeff.v:49-65   t1 10 b | t0 13 b as 2^12 - t0 | s 3/4 b as eta - s |
//                                               z 18/20 b as gamma1 - z        (-> canonical [0,q))
//   encoder.v:96-133                            the inverse maps (w1 packing lives in hash_kernels.hip)
//   usehint.v:92-114 / makehint.v:104-150       hint = omega position bytes + K cumulative counts
//   gen_s.v, sampler_s.v, rejection_s.v:...     ExpandS: SHAKE256(rho' || LE16 nonce), nibble rejection
// Buffers may sit at any byte alignment and stride (signatures are 3293 bytes at level 3), so the
// codecs move bytes; they are a few KB per item next to the 30-56 KiB matrices.
#include "keccak.hpp"
#include "kernels.hpp"
#include "modarith.hpp"
#include "sampler_bodies.hpp"
#include "coop_bodies.hpp"

namespace dil {

constexpr int32_t QC = 8380417;

// XF_PLAIN: value = v   |   XF_OFFSET_MINUS: value = OFFSET - v   (kernels.hpp)

// ---------------------------------------------------------------------------------------
// unpack: out[item][poly][i] = canon(xf(bits [i*BITS, (i+1)*BITS) of the poly's stream))
// ---------------------------------------------------------------------------------------
template <int BITS>
__device__ __forceinline__ uint32_t read_bits(const uint8_t* __restrict__ p, uint32_t bitpos)
{
    const uint32_t byte = bitpos >> 3, sh = bitpos & 7, nbytes = (sh + BITS + 7) >> 3;
    uint32_t v = p[byte];
    if (nbytes > 1) v |= (uint32_t)p[byte + 1] << 8;
    if (nbytes > 2) v |= (uint32_t)p[byte + 2] << 16;
    if (nbytes > 3) v |= (uint32_t)p[byte + 3] << 24;
    return (v >> sh) & ((1u << BITS) - 1);
}

template <int BITS>
__global__ __launch_bounds__(256) void unpack_kernel(int32_t* __restrict__ out, const uint8_t* __restrict__ in,
                                                     size_t in_stride, size_t in_offset, int polys, int xf, int32_t offset,
                                                     size_t nitems)
{
    const size_t total = nitems * (size_t)polys * 256;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t item = g / ((size_t)polys * 256);
        const uint32_t r = (uint32_t)(g % ((size_t)polys * 256)), poly = r >> 8, i = r & 255;
        const uint8_t* p = in + item * in_stride + in_offset + (size_t)poly * (32 * BITS);
        int32_t v = (int32_t)read_bits<BITS>(p, i * BITS);
        if (xf == XF_OFFSET_MINUS) v = offset - v;
        out[g] = v + ((v >> 31) & QC);
    }
}

// ---------------------------------------------------------------------------------------
// pack: one thread per OUTPUT byte; value_i = xf(centred(in[i])) as a BITS-bit field
// ---------------------------------------------------------------------------------------
// 8 coefficients make exactly BITS bytes: one thread per such group (32 per polynomial), two 16-B
// loads, fields assembled at compile-time bit positions in a 160-bit accumulator.
template <int BITS>
__global__ __launch_bounds__(256) void pack_kernel(uint8_t* __restrict__ out, size_t out_stride, size_t out_offset,
                                                   const int32_t* __restrict__ in, int polys, int xf, int32_t offset,
                                                   size_t nitems, RowMap map)
{
    if (map.count) nitems = min(nitems, (size_t)*map.count);
    const size_t total = nitems * (size_t)polys * 32;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t item = g / ((size_t)polys * 32);
        const uint32_t r = (uint32_t)(g % ((size_t)polys * 32)), poly = r >> 5, grp = r & 31;
        const size_t irow = map.src_row ? (size_t)map.src_row[item] : item, orow = map.dst_row ? (size_t)map.dst_row[item] : item;
        const int4* src = reinterpret_cast<const int4*>(in + (irow * polys + poly) * 256 + grp * 8);
        const int4 lo = src[0], hi = src[1];
        const int32_t c[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int32_t v = c[i] % QC;                                     // any representative
            v += (v >> 31) & QC;                                       // canonical
            v -= (((QC - 1) / 2 - v) >> 31) & QC;                      // centred
            const uint32_t f = (uint32_t)(xf == XF_OFFSET_MINUS ? offset - v : v) & ((1u << BITS) - 1);
            const int bit = i * BITS, wi = bit >> 5, sh = bit & 31;
            w[wi] |= f << sh;
            if (sh + BITS > 32) w[wi + 1] |= f >> (32 - sh);
        }
        uint8_t* dst = out + orow * out_stride + out_offset + (size_t)poly * (32 * BITS) + (size_t)grp * BITS;
#pragma unroll
        for (int bt = 0; bt < BITS; bt++) dst[bt] = (uint8_t)(w[bt >> 2] >> (8 * (bt & 3)));
    }
}

// ---------------------------------------------------------------------------------------
// End of key generation in ONE launch (large batches; was: tr = H(pk) on a helper stream beside two field copies and two
// pack launches, joined by events).  Workgroups of one wave, two roles:
//   [0, h_blocks)   tr = SHAKE256(pk, 32), two lanes per sponge (a 10 .. 20-permutation dependency chain per key), written
//                   straight into sk[64:96]; the same lanes copy rho and key from the seed expansion into sk[0:64]
//   the rest        s1 and s2 packed (eta - x, 3 | 4 bits) into sk[96:...], 8 coefficients per thread as pack_kernel
// pk / sk 4-byte aligned (the fused keygen path), e = rho(32) | rho'(64) | key(32) per key.
// ---------------------------------------------------------------------------------------
template <int EB>
__global__ __launch_bounds__(64) void keygen_finish_kernel(uint8_t* __restrict__ sk, size_t sk_bytes, const uint8_t* __restrict__ pk,
                                                           size_t pk_bytes, const uint8_t* __restrict__ e, const int32_t* __restrict__ s1,
                                                           const int32_t* __restrict__ s2, int L, int K, int32_t eta, unsigned h_blocks,
                                                           size_t nkeys, int coop_h)
{
    if (blockIdx.x < h_blocks && coop_h) {         // few keys: a key per workgroup, its sponge spread over the wave (coop_bodies.hpp)
        const size_t i = blockIdx.x;
        uint32_t* dst = reinterpret_cast<uint32_t*>(sk + i * sk_bytes);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(e + i * 128);
        if (threadIdx.x < 8) dst[threadIdx.x] = src[threadIdx.x];                    // rho
        else if (threadIdx.x < 16) dst[threadIdx.x] = src[16 + threadIdx.x];         // key
        coop::shake256_body(dst + 16, 4, reinterpret_cast<const uint32_t*>(pk + i * pk_bytes), (int)(pk_bytes / 8));
        return;
    }
    if (blockIdx.x < h_blocks) {
        const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
        const size_t i = t >> 1;
        if (i >= nkeys) return;                                   // whole pairs leave together
        const int hi = (int)(t & 1);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sk + i * sk_bytes);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(e + i * 128);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            dst[4 * hi + w] = src[4 * hi + w];                    // rho
            dst[8 + 4 * hi + w] = src[24 + 4 * hi + w];           // key
        }
        Shake2<17> sp;
        sp.init(hi);
        const int fill = sp.absorb<0>(reinterpret_cast<const uint32_t*>(pk + i * pk_bytes), (int)(pk_bytes / 8));
        sp.finish_words(fill);
        sp.squeeze(dst + 16, 4);
        return;
    }
    const size_t g = (size_t)(blockIdx.x - h_blocks) * 64 + threadIdx.x;
    const int polys = L + K;
    if (g >= nkeys * (size_t)polys * 32) return;
    const size_t item = g / ((size_t)polys * 32);
    const uint32_t r = (uint32_t)(g % ((size_t)polys * 32)), poly = r >> 5, grp = r & 31;
    const int32_t* base = poly < (uint32_t)L ? s1 + (item * L + poly) * 256 : s2 + (item * K + (poly - L)) * 256;
    const int4* src = reinterpret_cast<const int4*>(base + grp * 8);
    const int4 lo = src[0], hi4 = src[1];
    const int32_t c[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t v = c[i];                                          // ExpandS output: canonical
        v -= (((QC - 1) / 2 - v) >> 31) & QC;                      // centred
        w |= ((uint32_t)(eta - v) & ((1u << EB) - 1)) << (i * EB);
    }
    uint8_t* dst = sk + item * sk_bytes + 96 + (size_t)poly * (32 * EB) + (size_t)grp * EB;
#pragma unroll
    for (int bt = 0; bt < EB; bt++) dst[bt] = (uint8_t)(w >> (8 * bt));
}

// ---------------------------------------------------------------------------------------
// hints.  Wire form: omega position bytes then K cumulative counts (usehint.v:92-114).
// unpack: one wave per item -> h [K][256] bytes 0/1; bad[item] = 1 when the encoding is malformed
// (counts not monotone / > omega, positions not strictly increasing inside a row, non-zero padding).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hint_unpack_kernel(uint8_t* __restrict__ h, int32_t* __restrict__ bad,
                                                          const uint8_t* __restrict__ in, size_t in_stride, size_t in_offset,
                                                          int K, int omega, size_t nitems)
{
    const int lane = threadIdx.x & 63;
    const size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= nitems) return;
    const uint8_t* src = in + it * in_stride + in_offset;
    uint8_t* dst = h + it * (size_t)K * 256;
    for (int k = lane; k < K * 64; k += 64) reinterpret_cast<uint32_t*>(dst)[k] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    bool err = false;
    int prev_cnt = 0;
    for (int k = 0; k < K; k++) {
        const int cnt = src[omega + k];
        if (cnt < prev_cnt || cnt > omega) err = true;
        prev_cnt = cnt;
    }
    const int total = err ? 0 : prev_cnt;
    for (int t = lane; t < omega + ((64 - omega % 64) % 64); t += 64) {
        if (t < omega) {
            const int pos = src[t];
            if (t < total) {
                int row = 0, row_start = 0;
                for (int k = 0; k < K; k++) {
                    const int cnt = src[omega + k];
                    if (cnt <= t) { row = k + 1; row_start = cnt; }
                }
                if (t > row_start && pos <= (int)src[t - 1]) err = true;
                if (row < K) dst[row * 256 + pos] = 1;
            } else if (pos != 0) {
                err = true;
            }
        }
    }
    const bool any = __ballot(err) != 0;
    if (lane == 0) bad[it] = any ? 1 : 0;
}

// pack: h [K][256] bytes -> omega + K bytes (makehint.v:104-150).  One wave per item; a row is read as one dword per lane
// (coefficients 4 lane .. 4 lane + 3) and all K loads are issued before anything waits -- a byte per lane and chunk was a chain of
// 4 K dependent loads (10.5 -> ~6 us per signing round, profiles/r04u_ab_sign_finish.txt).  Positions ascend by coefficient.
__global__ __launch_bounds__(256) void hint_pack_kernel(uint8_t* __restrict__ out, size_t out_stride, size_t out_offset,
                                                        const uint8_t* __restrict__ h, int K, int omega, size_t nitems, RowMap map)
{
    const int lane = threadIdx.x & 63;
    const size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map.count) nitems = min(nitems, (size_t)*map.count);
    if (it >= nitems) return;
    const size_t irow = map.src_row ? (size_t)map.src_row[it] : it, orow = map.dst_row ? (size_t)map.dst_row[it] : it;
    uint8_t* dst = out + orow * out_stride + out_offset;
    const uint8_t* src = h + irow * (size_t)K * 256;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {                    // (K <= 8; a clamped row instead of a branch keeps the loads together)
        uint32_t v;
        __builtin_memcpy(&v, src + (k < K ? k : 0) * 256 + 4 * lane, 4);
        w[k] = v;
    }
    for (int t = lane; t < omega + K; t += 64) dst[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const unsigned long long lt = (1ull << lane) - 1;
    int count = 0;                                   // wave-uniform
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= K) break;
        bool nz[4];
        int below = 0, total = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            nz[b] = ((w[k] >> (8 * b)) & 0xFFu) != 0;
            const unsigned long long m = __ballot(nz[b]);
            below += __popcll(m & lt);
            total += __popcll(m);
        }
        int r = count + below;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (nz[b]) {
                if (r < omega) dst[r] = (uint8_t)(4 * lane + b);
                r++;
            }
        count += total;
        if (lane == 0) dst[omega + k] = (uint8_t)(count < omega ? count : omega);
    }
}

// ---------------------------------------------------------------------------------------
// ExpandS (gen_s.v / rejection_s.v): s[item][n] = RejEta(SHAKE256(rho' || LE16(nonce0 + n))),
// eta = 2: nibble < 15 -> 2 - (nibble mod 5);  eta = 4: nibble < 9 -> 4 - nibble.  Canonical out.
// One lane per polynomial.
// ---------------------------------------------------------------------------------------
// ExpandS, one lane per sponge (every batch size: measured faster than a two-lane form with per-nibble branches even for 100
// keys, 39 vs 59 us).  An accepted nibble costs six straight-line instructions: the RAW nibble is
// written to the lane's byte row in LDS at its running count (a rejected one is overwritten by the next), the count
// advances by the accept mask, and the `cnt < 256` guard is evaluated once per 64-bit word (a row has 16 spare bytes for
// the overshoot).  Only when every lane of the wave has its 256 nibbles does the wave turn to polynomial layout: lane t
// reads nibbles 4t..4t+3 of polynomial after polynomial, maps them to eta - (nibble [mod 5]) (rejection_s.v), and stores
// whole 1-KiB polynomials coalesced.  Row stride 69 dwords: conflict-free for both the byte writes (lanes at similar
// counts) and the dword reads.
template <int ETA>
__global__ __launch_bounds__(HASH_BS) void expand_s_fast_kernel(int32_t* __restrict__ s, int32_t* __restrict__ s_tail, int split,
                                                                const uint8_t* __restrict__ rhoprime, size_t rp_stride, int nonce0,
                                                                int polys, size_t nitems)
{
    __shared__ uint32_t buf[EXPAND_S_LDS_DWORDS];
    expand_s_fast_body<ETA>(s, s_tail, split, rhoprime, rp_stride, nonce0, polys, nitems, blockIdx.x, buf);
}

// t = w + s2 (mod q);  (t1, t0) = Power2Round(t), d = 13  (combined_top.v keygen :921-1079)
// t1 out as 10-bit values, t0 out centred in (-2^12, 2^12]
__global__ __launch_bounds__(256) void power2round_kernel(int32_t* __restrict__ t1, int32_t* __restrict__ t0,
                                                          const int32_t* __restrict__ w, const int32_t* __restrict__ s2, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int32_t t = (w[i] % QC + s2[i] % QC) % QC;
        t += (t >> 31) & QC;
        const int32_t hi = (t + (1 << 12) - 1) >> 13;
        t1[i] = hi;
        t0[i] = t - (hi << 13);
    }
}

// strided byte copy: dst[item][dst_off .. +n) = src[item][src_off .. +n)
__global__ __launch_bounds__(256) void copy_field_kernel(uint8_t* __restrict__ dst, size_t dst_stride, size_t dst_off,
                                                         const uint8_t* __restrict__ src, size_t src_stride, size_t src_off,
                                                         int nbytes, size_t nitems, RowMap map)
{
    if (map.count) nitems = min(nitems, (size_t)*map.count);
    const size_t total = nitems * (size_t)nbytes;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t item = g / (size_t)nbytes, b = g % (size_t)nbytes;
        const size_t irow = map.src_row ? (size_t)map.src_row[item] : item, orow = map.dst_row ? (size_t)map.dst_row[item] : item;
        dst[orow * dst_stride + dst_off + b] = src[irow * src_stride + src_off + b];
    }
}

// Rejection-loop bookkeeping (row N3).  A round works on E = n * S "entries": entry e is attempt
// number a0 + e % S of pending item e / S (item-major: the S attempts of one item are adjacent, so
// their reads of the item's key material hit L2).
//
// dst[e] = src[item(e / S)] for rows of `row_vec` VEC-sized words; item(i) = idx ? idx[i] : i
template <typename VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(VEC* __restrict__ dst, const VEC* __restrict__ src,
                                                          const int32_t* __restrict__ idx, size_t row_vec, uint32_t S, size_t entries)
{
    const size_t total = entries * row_vec;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        const size_t e = g / row_vec, w = g % row_vec, i = e / S;
        dst[g] = src[(size_t)(idx ? idx[i] : (int32_t)i) * row_vec + w];
    }
}

// kappa[e] = (a0 + e % S) * L      (the reference's y-nonce counter advances by L per attempt)
__global__ __launch_bounds__(256) void sign_kappa_kernel(uint32_t* __restrict__ kappa, int32_t* __restrict__ flags, uint32_t a0, uint32_t L,
                                                         uint32_t S, size_t entries)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < entries) {
        kappa[e] = (a0 + (uint32_t)(e % S)) * L;
        flags[e] = -1;
    }
}

// Start of a signing round in ONE launch (was two gathers, the kappa kernel and a memset): for entry e of pending item
// i = e / S:  mu_c[e] = mu[item(i)], rp_c[e] = rho'[item(i)] (64 bytes each, as 4 x 16 B per thread), kappa[e] = (a0 + e % S) * L,
// the entry's flag = -1 ("no verdict yet": phase 2 of a speculative round looks at the flags of an item's earlier attempts while it
// runs, pipelines.hip), and the round's two counters are cleared.  gather == false (first round of a full batch, S == 1): no gathers.
__global__ __launch_bounds__(256) void sign_round_setup_kernel(uint4* __restrict__ mu_c, uint4* __restrict__ rp_c,
                                                               uint32_t* __restrict__ kappa, int32_t* __restrict__ flags,
                                                               int32_t* __restrict__ counts, uint32_t* __restrict__ tickets,
                                                               const uint4* __restrict__ mu, const uint4* __restrict__ rp,
                                                               const int32_t* __restrict__ idx, uint32_t a0, uint32_t L, uint32_t S,
                                                               size_t entries, int gather)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < 3) counts[g] = 0;          // pending, winners, workgroups of the collect kernel that are through
    for (size_t i = g; i < (size_t)TICKET_WORDS; i += (size_t)gridDim.x * blockDim.x) tickets[i] = 0;      // phase 2's work queues (KeyMap::ticket)
    const size_t e = g >> 2, w = g & 3;
    if (e >= entries) return;
    if (w == 0) kappa[e] = (a0 + (uint32_t)(e % S)) * L;
    if (w == 1) flags[e] = -1;
    if (gather) {
        const size_t i = e / S, item = idx ? (size_t)idx[i] : i;
        mu_c[g] = mu[item * 4 + w];
        rp_c[g] = rp[item * 4 + w];
    }
}

// End of a signing round, one thread per pending item: the FIRST accepted of its S speculative attempts wins -> the
// (entry, item) pair goes on the winners list (packed into the item's signature slot by the RowMap-driven codec launches
// that follow), the attempt count is recorded and the winner's c~ (32 bytes, any alignment) is copied into its signature
// slot; none accepted -> the item goes on the next pending list.  counts[0] = pending, counts[1] = winners.
// host_words != nullptr: the LAST workgroup through (counts[2]) posts the two counts and then `seq` into the host's mapped, coherent
// words -- the host sizes the next round from them the moment this kernel is done, without a copy, an event or a wake-up in between
// (scheme.hip sign_core; the FPGA's FSM never leaves the device between attempts either, combined_top.v:1823-1934).
__global__ __launch_bounds__(256) void sign_collect_ct_kernel(int32_t* __restrict__ attempts, int32_t* __restrict__ next_idx,
                                                              int32_t* __restrict__ win_entry, int32_t* __restrict__ win_item,
                                                              int32_t* __restrict__ counts, const int32_t* __restrict__ flags,
                                                              const int32_t* __restrict__ idx, int a0, int S, size_t n,
                                                              uint8_t* __restrict__ sig, size_t sig_stride, const uint8_t* __restrict__ ct,
                                                              int32_t* host_words, uint32_t seq)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t item = idx ? idx[i] : (int32_t)i;
        int win = -1;
        for (int j = 0; j < S; j++)
            if (flags[i * (size_t)S + j] == 0) {
                win = j;
                break;
            }
        if (win < 0) {
            next_idx[atomicAdd(&counts[0], 1)] = item;
        } else {
            const int w = atomicAdd(&counts[1], 1);
            const size_t entry = i * (size_t)S + win;
            win_entry[w] = (int32_t)entry;
            win_item[w] = item;
            attempts[item] = a0 + win + 1;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ct + entry * 32);      // scratch: 4-byte aligned
            uint8_t* dst = sig + (size_t)item * sig_stride;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t v = src[q];
                __builtin_memcpy(dst + 4 * q, &v, 4);
            }
        }
    }
    if (!host_words) return;
    __syncthreads();                                     // this workgroup's additions to counts[0..1] are issued
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&counts[2], 1) == (int)gridDim.x - 1) {
            __threadfence();
            const int32_t pending = __hip_atomic_load(&counts[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int32_t winners = __hip_atomic_load(&counts[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&host_words[0], pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_words[1], winners, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<uint32_t*>(&host_words[2]), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// verdict[i] |= flag[i] ? bit : 0
__global__ __launch_bounds__(256) void or_flag_kernel(int32_t* __restrict__ verdict, const int32_t* __restrict__ flag, int bit, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) verdict[i] |= bit;
}

// dil_sign_msg_dev: an item whose message reference left the blob gets no signature (zero bytes, attempts = -1); one wave per item
__global__ __launch_bounds__(256) void sign_void_bad_kernel(uint8_t* __restrict__ sig, size_t sig_bytes, int32_t* __restrict__ attempts,
                                                            const int32_t* __restrict__ bad, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n || !bad[i]) return;
    for (size_t b = threadIdx.x & 63; b < sig_bytes; b += 64) sig[i * sig_bytes + b] = 0;
    if ((threadIdx.x & 63) == 0 && attempts) attempts[i] = -1;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static inline int grid1d(size_t n, const Tables& t)
{
    const size_t blocks = (n + 255) / 256, cap = (size_t)t.num_cus * 8;
    return (int)(blocks < 1 ? 1 : (blocks < cap ? blocks : cap));
}

hipError_t launch_or_flag(int32_t* verdict, const int32_t* flag, int bit, size_t n, const Tables& t, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(or_flag_kernel, (int)((n + 255) / 256), 256, 0, s, verdict, flag, bit, n);
    return hipGetLastError();
}

hipError_t launch_sign_void_bad(uint8_t* sig, size_t sig_bytes, int32_t* attempts, const int32_t* bad, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sign_void_bad_kernel, (int)((n + 3) / 4), 256, 0, s, sig, sig_bytes, attempts, bad, n);
    return hipGetLastError();
}

hipError_t launch_gather_rows(void* dst, const void* src, const int32_t* idx, size_t row_bytes, uint32_t S, size_t entries,
                              const Tables& t, hipStream_t s)
{
    if (entries == 0) return hipSuccess;
    const uintptr_t al = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | row_bytes;
    if (al % 16 == 0) {
        const size_t rv = row_bytes / 16;
        hipLaunchKernelGGL(gather_rows_kernel<uint4>, grid1d(entries * rv, t), 256, 0, s, static_cast<uint4*>(dst),
                           static_cast<const uint4*>(src), idx, rv, S, entries);
    } else if (al % 8 == 0) {
        const size_t rv = row_bytes / 8;
        hipLaunchKernelGGL(gather_rows_kernel<uint2>, grid1d(entries * rv, t), 256, 0, s, static_cast<uint2*>(dst),
                           static_cast<const uint2*>(src), idx, rv, S, entries);
    } else {
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, grid1d(entries * row_bytes, t), 256, 0, s, static_cast<uint8_t*>(dst),
                           static_cast<const uint8_t*>(src), idx, row_bytes, S, entries);
    }
    return hipGetLastError();
}

hipError_t launch_sign_kappa(uint32_t* kappa, int32_t* flags, uint32_t a0, uint32_t L, uint32_t S, size_t entries, hipStream_t s)
{
    if (entries == 0) return hipSuccess;
    hipLaunchKernelGGL(sign_kappa_kernel, (int)((entries + 255) / 256), 256, 0, s, kappa, flags, a0, L, S, entries);
    return hipGetLastError();
}

hipError_t launch_sign_round_setup(uint8_t* mu_c, uint8_t* rp_c, uint32_t* kappa, int32_t* flags, int32_t* counts, uint32_t* tickets,
                                   const uint8_t* mu, const uint8_t* rp,
                                   const int32_t* idx, uint32_t a0, uint32_t L, uint32_t S, size_t entries, bool gather, hipStream_t s)
{
    if (entries == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(mu_c) | reinterpret_cast<uintptr_t>(rp_c) | reinterpret_cast<uintptr_t>(mu) |
         reinterpret_cast<uintptr_t>(rp)) & 15)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(sign_round_setup_kernel, (int)((entries * 4 + 255) / 256), 256, 0, s, reinterpret_cast<uint4*>(mu_c),
                       reinterpret_cast<uint4*>(rp_c), kappa, flags, counts, tickets, reinterpret_cast<const uint4*>(mu),
                       reinterpret_cast<const uint4*>(rp),
                       idx, a0, L, S, entries, gather ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_sign_collect_ct(int32_t* attempts, int32_t* next_idx, int32_t* win_entry, int32_t* win_item, int32_t* counts,
                                  const int32_t* flags, const int32_t* idx, int a0, int S, size_t n, uint8_t* sig, size_t sig_stride,
                                  const uint8_t* ct, hipStream_t s, int32_t* host_words, uint32_t seq)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sign_collect_ct_kernel, (int)((n + 255) / 256), 256, 0, s, attempts, next_idx, win_entry, win_item, counts, flags,
                       idx, a0, S, n, sig, sig_stride, ct, host_words, seq);
    return hipGetLastError();
}

hipError_t launch_unpack(int bits, int32_t* out, const uint8_t* in, size_t in_stride, size_t in_offset, int polys, int xf,
                         int32_t offset, size_t nitems, const Tables& t, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    const int g = grid1d(nitems * (size_t)polys * 256, t);
#define DIL_UP(B) case B: hipLaunchKernelGGL(unpack_kernel<B>, g, 256, 0, s, out, in, in_stride, in_offset, polys, xf, offset, nitems); break
    switch (bits) {
        DIL_UP(3); DIL_UP(4); DIL_UP(6); DIL_UP(10); DIL_UP(13); DIL_UP(18); DIL_UP(20);
    default: return hipErrorInvalidValue;
    }
#undef DIL_UP
    return hipGetLastError();
}

hipError_t launch_pack(int bits, uint8_t* out, size_t out_stride, size_t out_offset, const int32_t* in, int polys, int xf,
                       int32_t offset, size_t nitems, const Tables& t, hipStream_t s, RowMap map)
{
    if (nitems == 0) return hipSuccess;
    const int g = grid1d(nitems * (size_t)polys * 32, t);
#define DIL_PK(B) case B: hipLaunchKernelGGL(pack_kernel<B>, g, 256, 0, s, out, out_stride, out_offset, in, polys, xf, offset, nitems, map); break
    switch (bits) {
        DIL_PK(3); DIL_PK(4); DIL_PK(6); DIL_PK(10); DIL_PK(13); DIL_PK(18); DIL_PK(20);
    default: return hipErrorInvalidValue;
    }
#undef DIL_PK
    return hipGetLastError();
}

hipError_t launch_keygen_finish(uint8_t* sk, size_t sk_bytes, const uint8_t* pk, size_t pk_bytes, const uint8_t* e, const int32_t* s1,
                                const int32_t* s2, int L, int K, int eta, int eta_bits, size_t nkeys, hipStream_t s)
{
    if (nkeys == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(sk) | reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(e) | sk_bytes | pk_bytes) & 3)
        return hipErrorInvalidValue;
    const int coop_h = coop_wanted(nkeys);
    const unsigned h_blocks = coop_h ? (unsigned)nkeys : (unsigned)((2 * nkeys + 63) / 64);
    const size_t p_blocks = (nkeys * (size_t)(L + K) * 32 + 63) / 64;
    if (h_blocks + p_blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(h_blocks + p_blocks);
    if (eta_bits == 3) hipLaunchKernelGGL(keygen_finish_kernel<3>, grid, 64, 0, s, sk, sk_bytes, pk, pk_bytes, e, s1, s2, L, K, eta, h_blocks, nkeys, coop_h);
    else if (eta_bits == 4) hipLaunchKernelGGL(keygen_finish_kernel<4>, grid, 64, 0, s, sk, sk_bytes, pk, pk_bytes, e, s1, s2, L, K, eta, h_blocks,
             nkeys, coop_h);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_hint_unpack(uint8_t* h, int32_t* bad, const uint8_t* in, size_t in_stride, size_t in_offset, int K, int omega,
                              size_t nitems, hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    hipLaunchKernelGGL(hint_unpack_kernel, (int)((nitems + 3) / 4), 256, 0, s, h, bad, in, in_stride, in_offset, K, omega, nitems);
    return hipGetLastError();
}

hipError_t launch_hint_pack(uint8_t* out, size_t out_stride, size_t out_offset, const uint8_t* h, int K, int omega, size_t nitems,
                            hipStream_t s, RowMap map)
{
    if (nitems == 0) return hipSuccess;
    hipLaunchKernelGGL(hint_pack_kernel, (int)((nitems + 3) / 4), 256, 0, s, out, out_stride, out_offset, h, K, omega, nitems, map);
    return hipGetLastError();
}

hipError_t launch_expand_s(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride, int eta, int L, int K, size_t nitems,
                           hipStream_t s)
{
    if (nitems == 0) return hipSuccess;
    const size_t total = nitems * (size_t)(L + K);
    if (coop_wanted(total)) return launch_coop_expand_s(s1, s2, rhoprime, rp_stride, eta, L, K, nitems, s);
    if (eta == 2)
        hipLaunchKernelGGL(expand_s_fast_kernel<2>, (int)((total + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, s1, s2, L, rhoprime, rp_stride, 0,
                           L + K, nitems);
    else
        hipLaunchKernelGGL(expand_s_fast_kernel<4>, (int)((total + HASH_BS - 1) / HASH_BS), HASH_BS, 0, s, s1, s2, L, rhoprime, rp_stride, 0,
                           L + K, nitems);
    return hipGetLastError();
}

hipError_t launch_power2round(int32_t* t1, int32_t* t0, const int32_t* w, const int32_t* s2, size_t n, const Tables& t, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(power2round_kernel, grid1d(n, t), 256, 0, s, t1, t0, w, s2, n);
    return hipGetLastError();
}

hipError_t launch_copy_field(uint8_t* dst, size_t dst_stride, size_t dst_off, const uint8_t* src, size_t src_stride, size_t src_off,
                             int nbytes, size_t nitems, const Tables& t, hipStream_t s, RowMap map)
{
    if (nitems == 0 || nbytes == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_field_kernel, grid1d(nitems * (size_t)nbytes, t), 256, 0, s, dst, dst_stride, dst_off, src, src_stride,
                       src_off, nbytes, nitems, map);
    return hipGetLastError();
}

}  // namespace dil
